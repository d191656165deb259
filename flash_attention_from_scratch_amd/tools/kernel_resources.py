#!/usr/bin/env python3
"""VGPR / AGPR / LDS / spill report of every device variant, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks -- the counterpart of the reference's
ptxas-log scraper (tools/build/parse_ptx_build.py -> regs / spills CSV).  No GPU needed.

    kernel_resources.py [--csv out.csv]
"""
import argparse
import csv
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "csrc")
SLICES = [("fa_inst.hip", 15, 1), ("fa_inst.hip", 15, 2), ("fa_inst.hip", 5, 1), ("fa_inst.hip", 5, 2),
          ("fa_inst16.hip", 15, 0), ("fa_inst16.hip", 5, 0)]
FIELDS = {"VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch_bytes",
          "Occupancy [waves/SIMD]": "occupancy", "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill"}


def demangle_variant(name):
    """_ZN2fa13fa_fwd_kernelILi15ELi1ELi8ELi64ELb1ELb1ELb0ELb1ELi0EEE... -> dict"""
    nums = re.findall(r"L[ib](\d+)E", name)
    if "fa_fwd_kernel64" in name and len(nums) >= 2:  # <DT, MASK, ABL, RAG, SPEC>
        dt, masked = map(int, nums[:2])
        rag = int(nums[3]) if len(nums) >= 4 else 0  # masked: 2 = causal form, 3 = ragged form of the same entry
        spec = int(nums[4]) if len(nums) >= 5 else 0  # the speculative softmax = optimized_softmax
        qtp = int(nums[6]) if len(nums) >= 7 else 2   # 1: the ring form of (B_r 128, B_c 64, 4 waves) + buffer
        nw = int(nums[8]) if len(nums) >= 9 else 4    # <..., QTP, ALT, NW>: 8 = the eight-wave ring form (round 6)
        return dict(dtype=dt, rows_per_wave=32 * qtp, n_waves=nw, B_c=64, swizzled=1, eager=1, opt_softmax=spec,
                    pipelined=1, dma=1, masked=2 * masked + rag, d_head=128, ring_kernel=1)
    if "fa_fwd_kernel16" in name and len(nums) >= 6:
        dt, nw, bc, swz, eager, opt = map(int, nums[:6])
        return dict(dtype=dt, rows_per_wave=16, n_waves=nw, B_c=bc, swizzled=swz, eager=eager,
                    opt_softmax=opt, pipelined=0, dma=1, masked=0, d_head=128)
    if len(nums) >= 11:
        dt, qt, nw, bc, swz, eager, opt, pipe, dma, masked, d_head = map(int, nums[:11])
        ksplit = int(nums[12]) if len(nums) >= 13 else 1  # <..., ABL, KSPLIT>: KSPLIT waves share a 32-row group (B_r = rows * waves)
        return dict(dtype=dt, rows_per_wave=32 * qt // ksplit, n_waves=nw, B_c=bc, swizzled=swz, eager=eager,
                    opt_softmax=opt, pipelined=pipe, dma=dma, masked=masked, d_head=d_head)
    return {}


def parse_remarks(text):
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark: .*?Function Name: (\S+)", line)
        if m:
            cur = {"kernel": m.group(1)}
            cur.update(demangle_variant(m.group(1)))
            rows.append(cur)
            continue
        if "remark" not in line or cur is None:
            continue
        for label, key in FIELDS.items():
            m = re.search(r"(?<![A-Za-z] )" + re.escape(label) + r": (\d+)", line)
            if m and line.split(label)[0].rstrip().endswith((":", "remark:")):
                cur[key] = int(m.group(1))
    return rows


def collect():
    rows = []
    for src, dt, qt in SLICES:
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-DFA_INST_DT={dt}",
               f"-DFA_INST_QT={qt}", "-Rpass-analysis=kernel-resource-usage", "-I", CSRC, "-c",
               os.path.join(CSRC, src), "-o", os.devnull]
        if qt == 2:
            cmd.insert(1, "-fno-slp-vectorize")  # as csrc/Makefile builds the 64-rows-per-wave slice (QT2FLAGS)
        rows += parse_remarks(subprocess.run(cmd, capture_output=True, text=True).stderr)
    for r in rows:
        r["B_r"] = r.get("rows_per_wave", 0) * r.get("n_waves", 0)
        stages = 2 if r.get("eager") else 1
        r["lds_bytes"] = max(2 * stages * r.get("B_c", 0), r["B_r"]) * 2 * r.get("d_head", 128)
        if r.get("ring_kernel"):
            # the persistent schedule (fa_fwd_kernel64, either number of Q tiles per wave): 4-stage K and V rings + 8 KiB of
            # staging per wave (RingTraits::kLdsBytes)
            r["lds_bytes"] = (2 * 4 * r.get("B_c", 0) + 32 * r.get("n_waves", 0)) * 2 * r.get("d_head", 128)
    return rows


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--csv")
    args = ap.parse_args(argv)
    rows = collect()
    cols = ["dtype", "d_head", "B_r", "B_c", "n_waves", "rows_per_wave", "pipelined", "dma", "masked", "opt_softmax",
            "swizzled", "eager",
            "vgprs", "agprs", "sgprs", "sgpr_spill", "scratch_bytes", "vgpr_spill", "occupancy", "lds_bytes"]
    out = open(args.csv, "w", newline="") if args.csv else sys.stdout
    w = csv.writer(out)
    w.writerow(cols)
    for r in sorted(rows, key=lambda r: [r.get(c, 0) for c in cols[:12]]):
        w.writerow([r.get(c, "") for c in cols])
    spilled = [r["kernel"] for r in rows if r.get("scratch_bytes") or r.get("vgpr_spill")]
    if spilled:
        print(f"WARNING: {len(spilled)} variants spill", file=sys.stderr)
    return rows


if __name__ == "__main__":
    main()
