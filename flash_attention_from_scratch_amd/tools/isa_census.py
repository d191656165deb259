#!/usr/bin/env python3
"""Census of the HOT block of the persistent kernel (the four steady-state visits hipcc emits as one basic block of 256
MFMAs): every instruction by what it is there for (VERDICT r05 task 4).  Reads the ISA the build keeps
(csrc/build/qt2_dt15/fa_inst-*.s); prints per-visit counts.

    isa_census.py [--kernel REGEX] [--isa FILE]

unit work   = the softmax units the arithmetic needs: 64 v_fmamk (s c - m c), 64 v_exp, 64 v_add (row sums), 32 v_cvt_pk
operands    = 16 ds_read_b128 (K) + 32 ds_read_b64_tr_b16 (V^T) and their 16 counted waits
requests    = 8 LDS-DMA pieces, their 8 M0 writes, the request-pointer chain, the counted vmcnt wait + the barrier
other       = everything else"""
import argparse
import collections
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_ISA = os.path.join(HERE, "..", "csrc", "build", "qt2_dt15", "fa_inst-hip-amdgcn-amd-amdhsa-gfx950.s")
DEFAULT_KERNEL = r"_ZN2fa15fa_fwd_kernel64ILi15ELb0ELi0ELb0ELb1ELb0ELi2ELb0ELi4EEE"   # bf16, plain, speculative, 64 rows per wave


def blocks_of(text, kernel_regex):
    m = re.search(r"^(%s\w*):" % kernel_regex, text, re.M)
    if not m:
        raise SystemExit("kernel not found: " + kernel_regex)
    end = text.index(".Lfunc_end", m.end())
    cur, label, out = [], "entry", []
    for ln in text[m.end():end].splitlines():
        lab = re.match(r"^(\.LBB\d+_\d+):", ln)
        if lab:
            out.append((label, cur))
            cur, label = [], lab.group(1)
            continue
        t = ln.split(";")[0].strip()
        if t and not t.startswith("."):
            cur.append(t)
    out.append((label, cur))
    return out


def classify(ins):
    op = ins.split()[0]
    if "mfma" in op:
        return "MFMA"
    if op in ("v_fmamk_f32", "v_exp_f32_e32", "v_cvt_pk_bf16_f32", "v_cvt_pk_f16_f32", "v_cvt_pkrtz_f16_f32", "v_add_f32_e32"):
        return "unit work"
    if op.startswith("ds_read"):
        return "operands: LDS reads"
    if op == "s_waitcnt" and "lgkmcnt" in ins and "vmcnt" not in ins:
        return "operands: counted waits"
    if op.startswith("global_load_lds"):
        return "requests: DMA pieces"
    if op == "s_mov_b32" and ins.split()[1].startswith("m0"):
        return "requests: M0 (LDS destination)"
    if op in ("s_add_u32", "s_addc_u32"):
        return "requests: pointer chain"
    if op == "s_waitcnt" or op == "s_barrier":
        return "requests: counted wait + barrier"
    if op == "s_nop":
        return "other: hazard pads"
    return "other: guard checkpoint, loop"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--isa", default=DEFAULT_ISA)
    ap.add_argument("--kernel", default=DEFAULT_KERNEL)
    a = ap.parse_args()
    text = open(a.isa).read()
    hot = [(lab, ins) for lab, ins in blocks_of(text, a.kernel) if sum("mfma" in x.split()[0] for x in ins) >= 200]
    if len(hot) != 1:
        raise SystemExit("expected ONE block of >= 200 MFMAs, found %d" % len(hot))
    lab, ins = hot[0]
    n_mfma = sum("mfma" in x.split()[0] for x in ins)
    visits = n_mfma / 64.0
    by = collections.Counter(classify(x) for x in ins)
    ops = collections.defaultdict(collections.Counter)
    for x in ins:
        ops[classify(x)][x.split()[0]] += 1
    print("hot block %s: %d instructions, %d MFMAs = %.0f visits -> %.2f instructions per visit, %.2f per MFMA" %
          (lab, len(ins), n_mfma, visits, len(ins) / visits, len(ins) / n_mfma))
    for k in sorted(by, key=lambda k: (k.split(":")[0] != "MFMA", k)):
        detail = ", ".join("%s %.2f" % (o, c / visits) for o, c in ops[k].most_common())
        print("  %-34s %7.2f per visit   (%s)" % (k, by[k] / visits, detail))
    other = sum(c for k, c in by.items() if k != "MFMA" and k != "unit work" and not k.startswith("operands: LDS") and not k.startswith("requests: DMA"))
    print("  -> not MFMA / unit work / LDS read / DMA piece: %.2f per visit" % (other / visits))
    return 0


if __name__ == "__main__":
    sys.exit(main())
