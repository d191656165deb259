#!/usr/bin/env python3
"""Profiler-driven kernel table / autotune -- the MI355X counterpart of the reference's
tools/benchmark/ncu_bench.py + benchmark_autotune.sh (Nsight Compute CSV scrape).

For each seq_len it runs  rocprofv3 --kernel-trace [--pmc ...] -- run_kernels.py  as a
subprocess (the process boundary the reference has around `ncu`), parses rocprofv3's CSVs
and prints one table per seq_len sorted by duration:

    kernel (short form) | Dur (ms) | ratio | Cycles | VGPRs | AGPRs | LDS | scratch | L2 hit % | TFLOP/s

    KERNELS=tune   rocprof_bench.py --seq_lens 4096            # autotune sweep (benchmark_autotune.sh)
    KERNELS=native rocprof_bench.py --seq_lens 512,4096 --pmc  # + cycles and L2 hit rate

Counter passes use --kernel-trace + --pmc only (never sys/hip/hsa traces).  Results are also
appended to profiles/local_profiles/profile_<n>.csv with the git commit, as the reference does
(ncu_bench.py:416-434).
"""
import argparse
import collections
import csv
import glob
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

from flash_helpers.kernel_configs import (  # noqa: E402
    DType,
    FlashForwardKernelConfig,
    NativeKernelConfig,
    calc_mfma_flop,
    calc_self_attn_flop,
)
from flash_helpers.test.utils import BATCH_SIZE_FOR_SEQ_LEN, BENCHMARK_N_HEADS  # noqa: E402

PMC_GROUPS = [["GRBM_GUI_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum"]]


def symbol_to_config(symbol):
    """'void fa::fa_fwd_kernel<15, 1, 8, 64, true, true, false, true, true, false, 128, 0>(fa::KernelArgs)' ->
    config of that device variant (None for other kernels).  A variant whose OPT template flag builds the speculative
    softmax (fa_registry.hpp softmax_mode_of) comes back as a NativeKernelConfig with speculative_softmax set; where OPT is
    the reference's first-block skip, as optimized_softmax."""
    m = re.search(r"fa::fa_fwd_kernel(16|64)?<([^>]*)>", symbol)
    if not m:
        return None
    vals = [{"true": 1, "false": 0}.get(t.strip(), t.strip()) for t in m.group(2).split(",")]
    vals = [int(v) for v in vals]
    if m.group(1) == "64":  # fa_fwd_kernel64<DT, MASK, ABL, RAG, SPEC, PSQ, QTP, ALT>: the persistent (256, 64, 4) + buffer kernel,
        # or (QTP = 1) the ring form of (128, 64, 4) + buffer
        spec = bool(vals[4]) if len(vals) > 4 else False
        qtp = vals[6] if len(vals) > 6 else 2
        base = (DType(vals[0]), 128, 128 * qtp, 64, 4, True, True, True, 0, 0, 0, True, False)
        return NativeKernelConfig(*base, speculative_softmax=True) if spec else FlashForwardKernelConfig(*base)
    if m.group(1):
        dt, nw, bc, swz, eager, opt = vals[:6]
        rows, pipe = 16, 0
    else:
        dt, qt, nw, bc, swz, eager, opt, pipe, dma = vals[:9]
        d_head = vals[10] if len(vals) > 10 else 128
        ksplit = vals[12] if len(vals) > 12 else 1  # key split: KSPLIT waves share a 32-row group
        rows = 32 * qt // ksplit
        masked = bool(vals[9]) if len(vals) > 9 else False
        base = (DType(dt), d_head, rows * nw, bc, nw, bool(dma), bool(eager), bool(swz), 0, 0, 0, bool(pipe))
        if opt and eager and dma and not masked:
            return NativeKernelConfig(*base, False, speculative_softmax=True)
        return FlashForwardKernelConfig(*base, bool(opt))
    base = (DType(dt), 128, rows * nw, bc, nw, True, bool(eager), bool(swz), 0, 0, 0, bool(pipe))
    if opt and eager:
        return NativeKernelConfig(*base, False, speculative_softmax=True)
    return FlashForwardKernelConfig(*base, bool(opt))


def lds_of(cfg):
    try:
        from flash_attention_from_scratch_amd import _capi

        return _capi.lds_bytes(cfg) if cfg else None
    except Exception:
        return None


def git_commit():
    try:
        return subprocess.check_output(["git", "rev-parse", "HEAD"], cwd=ROOT, text=True).strip()
    except Exception:
        return "unknown"


def parse_kernel_trace(path):
    """-> {symbol: dict(durations_ns=[...], vgpr, agpr, lds, scratch)}"""
    out = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            if "fa_fwd_kernel" not in name:
                continue
            rec = out.setdefault(name, {"durations_ns": []})
            rec["durations_ns"].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            rec["vgpr"] = int(row.get("VGPR_Count", 0) or 0)
            rec["agpr"] = int(row.get("Accum_VGPR_Count", 0) or 0)
            rec["lds"] = int(row.get("LDS_Block_Size", 0) or 0)
            rec["scratch"] = int(row.get("Scratch_Size", 0) or 0)
    return out


def parse_counters(path):
    """-> {symbol: {counter: mean}} from a rocprofv3 counter_collection CSV."""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            if "fa_fwd_kernel" in row.get("Kernel_Name", ""):
                acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def run_rocprof(target_cmd, outdir, pmc=None):
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", outdir, "-o", "p"]
    if pmc:
        cmd += ["--pmc"] + pmc
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    subprocess.run(cmd + ["--"] + target_cmd, check=True, env=env, cwd="/tmp",
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def table_rows(trace, counters, batch, heads, seq_len, d_head, skip_first=1):
    rows = []
    for symbol, rec in trace.items():
        cfg = symbol_to_config(symbol)
        durs = rec["durations_ns"][skip_first:] or rec["durations_ns"]
        ms = sum(durs) / len(durs) / 1e6
        c = counters.get(symbol, {})
        hit, miss = c.get("TCC_HIT_sum"), c.get("TCC_MISS_sum")
        rows.append({
            "kernel": cfg.short_form() if cfg else symbol[:60],
            "dur_ms": ms,
            "cycles": c.get("GRBM_GUI_ACTIVE", 0) / 8 if c.get("GRBM_GUI_ACTIVE") else None,  # 8 XCDs
            # rocprofv3 reports only static LDS; these kernels use dynamic LDS -> ask the library
            "vgpr": rec.get("vgpr"), "agpr": rec.get("agpr"), "lds": rec.get("lds") or lds_of(cfg),
            "scratch": rec.get("scratch"),
            "l2_hit": 100 * hit / (hit + miss) if hit is not None and miss else None,
            "attn_tflops": calc_self_attn_flop(batch, heads, seq_len, d_head) / (ms * 1e-3) / 1e12,
            "mfma_tflops": calc_mfma_flop(batch, heads, seq_len, d_head) / (ms * 1e-3) / 1e12,
        })
    rows.sort(key=lambda r: r["dur_ms"])
    for r in rows:
        r["ratio"] = r["dur_ms"] / rows[0]["dur_ms"]
    return rows


def fmt(v, spec):
    return "" if v is None else format(v, spec)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--seq_lens", default="4096")
    ap.add_argument("--d_head", type=int, default=128)
    ap.add_argument("--n_runs", type=int, default=4)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--pmc", action="store_true", help="second pass: cycles + L2 hit rate")
    ap.add_argument("--kernels", nargs="+", help="explicit kernel strings (default: KERNELS env)")
    ap.add_argument("--no-log", action="store_true")
    args = ap.parse_args(argv)

    all_rows = []
    for seq_len in map(int, args.seq_lens.split(",")):
        batch = args.batch or BATCH_SIZE_FOR_SEQ_LEN[seq_len]
        target = [sys.executable, os.path.join(HERE, "run_kernels.py"), str(seq_len), str(args.d_head),
                  "--n_runs", str(args.n_runs), "--batch", str(batch)]
        if args.kernels:
            target += ["--kernels"] + args.kernels
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            run_rocprof(target, os.path.join(tmp, "t"))
            trace = parse_kernel_trace(glob.glob(os.path.join(tmp, "t", "**", "*kernel_trace.csv"), recursive=True)[0])
            counters = {}
            if args.pmc:
                for i, group in enumerate(PMC_GROUPS):
                    run_rocprof(target, os.path.join(tmp, f"c{i}"), group)
                    for f in glob.glob(os.path.join(tmp, f"c{i}", "**", "*counter_collection.csv"), recursive=True):
                        for k, v in parse_counters(f).items():
                            counters.setdefault(k, {}).update(v)
        rows = table_rows(trace, counters, batch, BENCHMARK_N_HEADS, seq_len, args.d_head)
        print(f"\nseq_len={seq_len} batch={batch} heads={BENCHMARK_N_HEADS} d_head={args.d_head}")
        print(f"{'Kernel':84s} {'Dur (ms)':>9s} {'ratio':>6s} {'Cycles':>9s} {'VGPR':>5s} {'AGPR':>5s} {'LDS':>6s} "
              f"{'scr':>4s} {'L2 hit%':>8s} {'TFLOP/s':>8s}")
        for r in rows:
            print(f"{r['kernel']:84s} {r['dur_ms']:9.4f} {r['ratio']:6.3f} {fmt(r['cycles'], '9.0f'):>9s} "
                  f"{fmt(r['vgpr'], 'd'):>5s} {fmt(r['agpr'], 'd'):>5s} {fmt(r['lds'], 'd'):>6s} "
                  f"{fmt(r['scratch'], 'd'):>4s} {fmt(r['l2_hit'], '8.2f'):>8s} {r['mfma_tflops']:8.1f}")
            all_rows.append(dict(r, seq_len=seq_len, batch=batch))
    if not args.no_log and all_rows:
        logdir = os.path.join(ROOT, "profiles", "local_profiles")
        os.makedirs(logdir, exist_ok=True)
        n = len(glob.glob(os.path.join(logdir, "profile_*.csv")))
        with open(os.path.join(logdir, f"profile_{n}.csv"), "w", newline="") as f:
            f.write(f"# git commit: {git_commit()}\n")
            w = csv.DictWriter(f, fieldnames=list(all_rows[0].keys()))
            w.writeheader()
            w.writerows(all_rows)
    return all_rows


if __name__ == "__main__":
    main()
