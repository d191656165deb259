#!/usr/bin/env python3
"""Toolchain pin of the hand-placed kernel: the opcode histogram of every steady-state VISIT of the persistent
64-rows-per-wave kernels, read from the ISA the build keeps (csrc/build/qt2_dt*/fa_inst-*.s), plus the hipcc version.

The visit's schedule lives at the edge of hipcc's register allocator: the MFMAs, DMA pieces and stores
are inline asm the compiler neither pads nor looks into, and a compiler upgrade re-rolls everything around them.  The
committed digest (profiles/r06/toolchain.json) is what the measured numbers of the round belong to; the CPU test
tests/test_tools_cpu.py::test_visit_histogram_matches_the_committed_digest fails when a rebuild no longer produces it.

    isa_digest.py [--write profiles/r06/toolchain.json]

A "visit" = a basic block of the kernel with >= 56 MFMAs (the barrier two MFMAs into a visit may split off a 3-MFMA
head, hence 61 or 64; since round 5 a walk is [first group | hot loop | last two groups] = twelve visit bodies, and hipcc
merges the branch-free hot ones into blocks of several visits: 256 MFMAs for the four of the hot loop).  Per visit the plan of DESIGN.md 3.5/3.6
deals 64 MFMAs, 64 v_exp_f32, 48 LDS operand reads (16 ds_read_b128 + 32 ds_read_b64_tr_b16) and 8 LDS-DMA pieces."""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "..", "csrc", "build")
OPS = ("mfma", "v_exp_f32", "v_fmamk_f32", "v_add_f32", "v_cvt_pk", "ds_read_b128", "ds_read_b64_tr_b16",
       "global_load_lds_dwordx4", "v_max3_f32", "v_readlane_b32", "v_writelane_b32", "v_accvgpr", "s_barrier")


def hipcc_version():
    try:
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True, check=True).stdout
    except (OSError, subprocess.SubprocessError):
        return "unknown"
    keep = [ln.strip() for ln in out.splitlines() if ln.startswith(("HIP version", "AMD clang version"))]
    return " | ".join(keep)


def visits_of(text, kernel_regex, lo=56, hi=1 << 30):
    """-> [{op class: count, "instructions": n}] for the visit blocks of the kernel matching `kernel_regex`: the basic blocks
    with lo <= MFMAs <= hi (the 64-row visits by default; lo, hi = 28, 32 gives the half-item visits of the round-6 second
    pass, one 32-row Q tile per wave = 16 + 16 MFMAs)."""
    m = re.search(r"^(_ZN2fa15fa_fwd_kernel64" + kernel_regex + r"EEvNS_10KernelArgsE):.*?\n(.*?)\n\s+s_endpgm", text, flags=re.S | re.M)
    if not m:
        return None
    blocks, cur = [], ("entry", [])
    for line in m.group(2).split("\n"):
        s = line.strip()
        lab = re.match(r"^(\.LBB\d+_\d+):", s)
        if lab:
            blocks.append(cur)
            cur = (lab.group(1), [])
            continue
        if not s or s.startswith((";", ".")):
            continue
        cur[1].append(s.split()[0])
    blocks.append(cur)
    out = []
    for _name, ops in blocks:
        c = collections.Counter()
        for op in ops:
            for cls in OPS:
                if (cls == "mfma" and op.startswith("v_mfma")) or (cls != "mfma" and op.startswith(cls)):
                    c[cls] += 1
                    break
        if lo <= c["mfma"] <= hi:
            out.append({"instructions": len(ops), **{k: c[k] for k in OPS}})
    return out


# (template arguments DT, MASK, ABL, RAG, SPEC, PSQ, QTP, ALT, NW as hipcc mangles them)
KERNELS = {
    "bf16 speculative (default)": ("qt2_dt15", "ILi15ELb0ELi0ELb0ELb1ELb0ELi2ELb0ELi4E"),
    "bf16 lazy": ("qt2_dt15", "ILi15ELb0ELi0ELb0ELb0ELb0ELi2ELb0ELi4E"),
    "fp16 speculative": ("qt2_dt5", "ILi5ELb0ELi0ELb0ELb1ELb0ELi2ELb0ELi4E"),
    "fp16 lazy (default)": ("qt2_dt5", "ILi5ELb0ELi0ELb0ELb0ELb0ELi2ELb0ELi4E"),
    "bf16 speculative, prescaled Q": ("qt2_dt15", "ILi15ELb0ELi0ELb0ELb1ELb1ELi2ELb0ELi4E"),
    "bf16 lazy, prescaled Q": ("qt2_dt15", "ILi15ELb0ELi0ELb0ELb0ELb1ELi2ELb0ELi4E"),
    "bf16 speculative, alternating K / V direction (long sequences)": ("qt2_dt15", "ILi15ELb0ELi0ELb0ELb1ELb0ELi2ELb1ELi4E"),
    "fp16 speculative, alternating K / V direction (long sequences)": ("qt2_dt5", "ILi5ELb0ELi0ELb0ELb1ELb0ELi2ELb1ELi4E"),
}


def digest():
    out = {"hipcc": hipcc_version(), "kernels": {}, "second_pass_half_visits": {}}
    cache = {}
    for name, (slice_dir, regex) in KERNELS.items():
        path = os.path.join(BUILD, slice_dir, "fa_inst-hip-amdgcn-amd-amdhsa-gfx950.s")
        if path not in cache:
            if not os.path.exists(path):
                return None
            cache[path] = open(path).read()
        out["kernels"][name] = visits_of(cache[path], regex)
        half = [v for v in (visits_of(cache[path], regex, 28, 32) or []) if v["v_max3_f32"] > 0]
        if half:
            out["second_pass_half_visits"][name] = half
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--write", default="", help="write the digest to this JSON file")
    args = ap.parse_args(argv)
    d = digest()
    if d is None:
        print("no kept ISA under csrc/build: run make -C flash_attention_from_scratch_amd/csrc first")
        return 1
    text = json.dumps(d, indent=1)
    if args.write:
        with open(args.write, "w") as f:
            f.write(text + "\n")
    print(text)
    return 0


if __name__ == "__main__":
    sys.exit(main())
