#!/usr/bin/env python3
"""Event-timed kernel comparison -- the MI355X counterpart of the reference's
tools/benchmark/pt_bench.py (its timing protocol :145-174, its statistics table :224-411).

Protocol kept from the reference: N warm-ups, then per repeat a cache flush (write a
buffer larger than L2 + Infinity Cache: 512 MiB here vs the reference's 100 MB for
A100's 40 MB L2, :36,98-99), an idle spin, and the in-extension event time of
flash_attention.forward_timed (preferred over outer events, :169-172).  Output is
the same CSV table (mean/median/min/max/stddev ms, % of the comparator, attention
TFLOP/s by calc_self_attn_flop) plus the roofline figure 4*B*H*S^2*d.  Clock
pinning (nvidia-smi -lgc, :111-134) is not done: the GPU pool runs every job at the
machine's default settings and refuses any command that would change them, so clocks
are READ (bench.py puts the hwmon sclk / power beside every number), never set.

Kernels are selected with the KERNELS env var exactly as in the reference
(kernel_configs.get_kernel_configs): all | tune | prog[all] | "B_r,B_c" | native | best.
"""
import argparse
import csv
import statistics
import sys

import torch

import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.abspath(_os.path.join(_os.path.dirname(__file__), _os.pardir, _os.pardir)))  # repo root: runs without PYTHONPATH

import flash_attention  # noqa: E402
from flash_helpers.kernel_configs import calc_mfma_flop, calc_self_attn_flop, get_kernel_configs
from flash_helpers.test.utils import (
    BATCH_SIZE_FOR_SEQ_LEN,
    BENCHMARK_N_HEADS,
    QKVConfig,
    generate_qkvo,
    reference_forward_kernel_v2_timed,
)

class Hermetic:
    """The between-repeat hygiene of the protocol: evict L2 + Infinity Cache by overwriting a 512-MiB
    buffer, then let the chip idle for a fixed spin so every repeat starts from the same state."""

    def __init__(self, device="cuda:0", nbytes=512 << 20, spin_cycles=1_000_000):
        self.scratch = torch.empty(nbytes, dtype=torch.int8, device=device)
        self.spin_cycles = spin_cycles

    def __call__(self):
        self.scratch.zero_()
        torch.cuda._sleep(self.spin_cycles)


class Timing:
    """Milliseconds of the repeats of one kernel, and what the table prints about them."""

    def __init__(self, ms):
        self.ms = sorted(ms)

    mean = property(lambda self: statistics.fmean(self.ms))
    median = property(lambda self: statistics.median(self.ms))
    min = property(lambda self: self.ms[0])
    max = property(lambda self: self.ms[-1])
    stddev = property(lambda self: statistics.stdev(self.ms) if len(self.ms) > 1 else 0.0)

    def tflops(self, flop):
        return flop / self.mean * 1e-9  # flop per ms -> TFLOP/s

    def percent_of(self, other):
        return 100.0 * other.mean / self.mean


_hygiene = None


@torch.inference_mode()
def time_kernel(kernel, warmups, repeats, hermetic=True) -> Timing:
    """`kernel()` returns (out, ms) -- the extension's own event bracket, preferred -- or just out, in
    which case a pair of events on the current stream times the call."""
    global _hygiene
    if hermetic and _hygiene is None:
        _hygiene = Hermetic()
    for _ in range(warmups):
        kernel()
    ms = []
    for _ in range(repeats):
        if hermetic:
            _hygiene()
        torch.cuda.synchronize()
        before, after = (torch.cuda.Event(enable_timing=True) for _ in range(2))
        before.record()
        result = kernel()
        after.record()
        after.synchronize()
        ms.append(result[1] if isinstance(result, tuple) else before.elapsed_time(after))
    return Timing(ms)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--d_heads", type=str, default="128")
    ap.add_argument("--seq_lens", type=str, default="512,1024,2048,4096")
    ap.add_argument("--num_warmups", type=int, default=10)
    ap.add_argument("--num_repeats", type=int, default=64)
    ap.add_argument("--noncu", action="store_true", help="no cache flush / idle spin between repeats")
    ap.add_argument("--batch", type=int, default=0, help="override BATCH_SIZE_FOR_SEQ_LEN")
    ap.add_argument("--heads", type=int, default=BENCHMARK_N_HEADS)
    ap.add_argument("--no-ref", action="store_true", help="skip the torch SDPA comparator")
    args = ap.parse_args(argv)

    device = torch.device("cuda:0")
    writer = csv.writer(sys.stdout)
    writer.writerow(["Kernel Name", "d_head", "seq_len", "batch", "Mean (ms)", "Median (ms)", "Min (ms)",
                     "Max (ms)", "StdDev (ms)", "Relative Performance", "Attn TFLOP/s",
                     "MFMA TFLOP/s (4BHS^2d)", "% of 2.5 PF peak"])
    harmonic = {}
    for d_head in map(int, args.d_heads.split(",")):
        for seq_len in map(int, args.seq_lens.split(",")):
            batch = args.batch or BATCH_SIZE_FOR_SEQ_LEN[seq_len]
            data = {}
            for dtype in (torch.float16, torch.bfloat16):
                data[dtype] = generate_qkvo(QKVConfig(n_heads=args.heads, d_head=d_head, batch_size=batch,
                                                      seq_len=seq_len, dtype=dtype, device=device), seed=0)
            attn_flops = calc_self_attn_flop(batch, args.heads, seq_len, d_head)
            mfma_flops = calc_mfma_flop(batch, args.heads, seq_len, d_head)
            rows = []
            ref_mean = None
            if not args.no_ref:
                q, k, v, o = data[torch.float16]
                ref = time_kernel(lambda: reference_forward_kernel_v2_timed(q, k, v, o),
                                  args.num_warmups, max(4, args.num_repeats // 4), not args.noncu)
                ref_mean = ref.mean
                rows.append(("Reference (torch SDPA fp16)", ref))
            for cfg in get_kernel_configs():
                if cfg.d_head != d_head or seq_len % cfg.B_r or seq_len % cfg.B_c:
                    continue
                q, k, v, o = data[cfg.dtype.to_torch_dtype()]
                st = time_kernel(lambda: flash_attention.forward_timed(kernel_cfg=cfg, q=q, k=k, v=v, o=o),
                                 args.num_warmups, args.num_repeats, not args.noncu)
                rows.append((cfg.short_form(), st))
                harmonic.setdefault(cfg.short_form(), []).append(st.tflops(mfma_flops))
            rows[1 if ref_mean else 0:] = sorted(rows[1 if ref_mean else 0:], key=lambda r: r[1].mean)
            for name, st in rows:
                rel = f"{st.percent_of(ref):.2f}%" if ref_mean else ""
                writer.writerow([name, d_head, seq_len, batch, f"{st.mean:.4f}", f"{st.median:.4f}",
                                 f"{st.min:.4f}", f"{st.max:.4f}", f"{st.stddev:.4f}", rel,
                                 f"{st.tflops(attn_flops):.2f}", f"{st.tflops(mfma_flops):.2f}",
                                 f"{100 * st.tflops(mfma_flops) / 2500:.1f}"])
            sys.stdout.flush()
    n_seq = len(args.seq_lens.split(",")) * len(args.d_heads.split(","))
    if n_seq > 1:
        writer.writerow([])
        writer.writerow(["Kernel Name", "harmonic-mean MFMA TFLOP/s over seq_lens"])
        for name, vals in sorted(harmonic.items(), key=lambda kv: -statistics.harmonic_mean(kv[1])):
            if len(vals) == n_seq:
                writer.writerow([name, f"{statistics.harmonic_mean(vals):.2f}"])


if __name__ == "__main__":
    main()
