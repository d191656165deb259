#!/usr/bin/env python3
"""fp16 vs bf16 at the C1 shape: is the gap cycles (the fp16 guard / convert path issues more, or stalls more) or clock
(the fp16 MFMA draws more power, so the chip clocks lower under the same cap)?  VERDICT r04 task 7.
Reads what tools/gpu_round.sh left in <dir>: pmc/pmc_summary.txt (bf16 kernels), pmc_fp16/pmc_summary.txt, bench_c1.json,
bench_c1_fp16.json, and prints one table.  Counter units: SQ_WAVE_CYCLES in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles
(MI355X_MICROARCH.md); one kernel launch = 16 842 752 MFMAs at C1.
Usage: python tools/fp16_vs_bf16.py gpurun_out/<tag>"""
import json
import os
import re
import sys


def parse_summary(path):
    out, cur = {}, None
    if not os.path.exists(path):
        return out
    for line in open(path):
        if line.startswith("void fa::"):
            cur = out.setdefault(line.strip(), {})
        else:
            m = re.match(r"\s+(\S+)\s+mean\s+([0-9.eE+-]+)", line)
            if m and cur is not None:
                cur[m.group(1)] = float(m.group(2))
    return out


def label(sym):
    m = re.search(r"fa_fwd_kernel64<(\d+), (\w+), \d+, (\w+), (\w+), (\w+)(?:, \d+)?(?:, \w+)?>", sym)
    if not m:
        return sym[:40]
    dt, _mask, _rag, spec, psq = m.groups()
    return f"{'fp16' if dt == '5' else 'bf16'} {'speculative' if spec == 'true' else 'lazy'}{' + pre-scaled Q' if psq == 'true' else ''}"


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else "."
    rows = {}
    for sub in ("pmc", "pmc_fp16"):
        for sym, c in parse_summary(os.path.join(d, sub, "pmc_summary.txt")).items():
            if "SQ_INSTS_MFMA" not in c:
                continue
            mf = c["SQ_INSTS_MFMA"]
            wave = 4.0 * c["SQ_WAVE_CYCLES"]
            rows[label(sym)] = {
                "wave cycles / MFMA": wave / mf,
                "VALU / MFMA": (c["SQ_INSTS_VALU"] - mf) / mf,
                "trans (v_exp) / MFMA": c.get("SQ_INSTS_VALU_TRANS", float("nan")) / mf,
                "LDS / MFMA": c["SQ_INSTS_LDS"] / mf,
                "SALU / MFMA": c["SQ_INSTS_SALU"] / mf,
                "MFMA busy / wave time": c.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")) / wave,
                "wait (s_waitcnt, barrier) / wave time": 4.0 * c.get("SQ_WAIT_ANY", float("nan")) / wave,
                "issue stall / wave time": 4.0 * c.get("SQ_WAIT_INST_ANY", float("nan")) / wave,
                "GRBM_GUI_ACTIVE per XCD (cycles)": c.get("GRBM_GUI_ACTIVE", float("nan")) / 8.0,
            }
    names = list(rows)
    if not names:
        print("no counter summaries under", d)
        return 1
    keys = list(rows[names[0]])
    w = max(len(k) for k in keys) + 2
    print("counters under rocprofv3 (profiled launches clock a few % lower than timed ones)")
    print(" " * w + "".join(f"{n:>28s}" for n in names))
    for k in keys:
        print(f"{k:<{w}s}" + "".join(f"{rows[n][k]:28.4f}" for n in names))
    print()
    print("timed, un-profiled (bench.py, the driver's protocol):")
    for tag, fn in (("bf16", "bench_c1.json"), ("fp16", "bench_c1_fp16.json")):
        p = os.path.join(d, fn)
        if not os.path.exists(p):
            continue
        r = json.load(open(p))
        clk = r.get("clocks", {})
        roof = r.get("roofline", {})
        pc = roof.get("pipe_counters") or {}
        print(f"  {tag}: {r['value']:.1f} TFLOP/s, kernel {roof.get('kernel_ms', float('nan')) * 1e3:.1f} us, sclk {clk.get('sclk_mhz', {}).get('mean')} MHz, "
              f"power {clk.get('power_w', {}).get('mean')} W of {clk.get('power_cap_w')} W, wave cycles / MFMA (in-run pass) {pc.get('wave_cycles_per_mfma')}, "
              f"MFMA-only roof on random data {roof.get('mfma_only_random_tflops')} TFLOP/s -> {roof.get('frac_of_mfma_only_random')} of it")
    b, f = None, None
    for n in names:
        if n == "bf16 speculative":
            b = rows[n]
        if n == "fp16 speculative":
            f = rows[n]
    if b and f:
        dc = f["wave cycles / MFMA"] / b["wave cycles / MFMA"] - 1.0
        print(f"\nfp16 vs bf16, speculative kernels: wave cycles per MFMA {dc * 100:+.2f} %; the rest of the TFLOP/s gap is clock "
              f"(cycles x clock = time: see the sclk and MFMA-only-roof columns above).")
    return 0


if __name__ == "__main__":
    sys.exit(main())
