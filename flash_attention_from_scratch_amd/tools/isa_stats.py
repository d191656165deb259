#!/usr/bin/env python3
"""gfx950 ISA statistics of the forward kernels -- the AMD counterpart of the
reference's SASS tooling (tools/analysis/count_sass_instructions.sh,
compare_sass_instruction_counts.py, tools/build/extract_sass.py).

    isa_stats.py build  --dtype 15 --qt 1 -o /tmp/fa.s      # hipcc -S of one variant slice
    isa_stats.py hist   /tmp/fa.s --kernel 'Li8ELi64E.*Lb1ELi0E' [--loop]
    isa_stats.py trace  /tmp/fa.s --kernel ... [--loop]      # one letter per instruction
    isa_stats.py diff   before.s after.s --kernel ...        # opcode count deltas

Classes: M mfma, t ds_read_b64_tr_b16, d other LDS, G LDS-DMA, g other VMEM, E v_exp
(transcendental), r v_pk_mul (O rescale), v other VALU, s SALU, <Ln>/<Vn> s_waitcnt,
|B| s_barrier, J branch.
"""
import argparse
import collections
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "csrc")


def build(dtype, qt, out, rows16=False):
    src = "fa_inst16.hip" if rows16 else "fa_inst.hip"
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", f"-DFA_INST_DT={dtype}",
           f"-DFA_INST_QT={qt}", "-S", "--cuda-device-only", "-I", CSRC, os.path.join(CSRC, src), "-o", out]
    subprocess.run(cmd, check=True)
    return out


def kernels(text):
    """-> {mangled name: [instruction lines]} for every kernel body in a .s file."""
    out = {}
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\n\s+s_endpgm", text, flags=re.S | re.M):
        out[m.group(1)] = m.group(2).split("\n")
    return out


def select(text, pattern):
    found = {k: v for k, v in kernels(text).items() if re.search(pattern, k)}
    if not found:
        raise SystemExit(f"no kernel matches {pattern!r}")
    return found


def hot_loop(lines):
    """Largest innermost loop: from its header label to the last backward branch to it."""
    best = None
    for i, line in enumerate(lines):
        if "Inner Loop Header" not in line:
            continue
        label = re.match(r"^(\.LBB\d+_\d+):", line)
        if not label:
            continue
        ends = [j for j, l in enumerate(lines) if j > i and re.search(r"s_c?branch\w*\s+" + re.escape(label.group(1)) + r"\b", l)]
        if ends and (best is None or ends[-1] - i > best[1] - best[0]):
            best = (i, ends[-1])
    return lines[best[0]:best[1] + 1] if best else lines


def opcode(line):
    m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)", line)
    return (m.group(1), m.group(2)) if m else (None, None)


def classify(op, rest=""):
    if op.startswith("v_mfma"):
        return "M"
    if op.startswith("ds_read_b64_tr"):
        return "t"
    if op.startswith("ds_"):
        return "d"
    if op.startswith("global_load_lds") or (op.startswith("buffer_load") and "lds" in rest):
        return "G"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "g"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt")):
        return "E"
    if op.startswith("v_pk_mul"):
        return "r"
    if op.startswith("v_"):
        return "v"
    if op == "s_waitcnt":
        return "<" + rest.split(")")[0].replace("lgkmcnt(", "L").replace("vmcnt(", "V") + ">"
    if op == "s_barrier":
        return "|B|"
    if op.startswith(("s_cbranch", "s_branch")):
        return "J"
    if op == "s_nop":
        return ""
    return "s"


def histogram(lines):
    ops = collections.Counter()
    for line in lines:
        op, _ = opcode(line)
        if op:
            ops[op] += 1
    return ops


def class_summary(ops):
    cls = collections.Counter()
    for op, n in ops.items():
        c = classify(op)
        key = {"M": "mfma", "t": "lds", "d": "lds", "G": "vmem", "g": "vmem", "E": "trans", "r": "valu",
               "v": "valu", "|B|": "barrier", "J": "branch", "s": "salu", "": "nop"}.get(c, "waitcnt")
        cls[key] += n
    return dict(cls)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    b = sub.add_parser("build")
    b.add_argument("--dtype", type=int, default=15)
    b.add_argument("--qt", type=int, default=1)
    b.add_argument("--rows16", action="store_true")
    b.add_argument("-o", "--out", required=True)
    for name in ("hist", "trace"):
        p = sub.add_parser(name)
        p.add_argument("asm")
        p.add_argument("--kernel", default=".")
        p.add_argument("--loop", action="store_true", help="restrict to the hot KV loop")
        p.add_argument("--top", type=int, default=40)
    d = sub.add_parser("diff")
    d.add_argument("before")
    d.add_argument("after")
    d.add_argument("--kernel", default=".")
    d.add_argument("--loop", action="store_true")
    args = ap.parse_args(argv)

    if args.cmd == "build":
        print(build(args.dtype, args.qt, args.out, args.rows16))
        return
    if args.cmd in ("hist", "trace"):
        for name, lines in select(open(args.asm).read(), args.kernel).items():
            body = hot_loop(lines) if args.loop else lines
            ops = histogram(body)
            print(f"{name}: {sum(ops.values())} instructions {class_summary(ops)}")
            if args.cmd == "hist":
                for op, n in ops.most_common(args.top):
                    print(f"{n:6d} {op}")
            else:
                seq = "".join(classify(*opcode(l)) for l in body if opcode(l)[0])
                for i in range(0, len(seq), 100):
                    print("  " + seq[i:i + 100])
        return
    before = select(open(args.before).read(), args.kernel)
    after = select(open(args.after).read(), args.kernel)
    for name in sorted(set(before) & set(after)):
        hb = histogram(hot_loop(before[name]) if args.loop else before[name])
        ha = histogram(hot_loop(after[name]) if args.loop else after[name])
        print(f"{name}: {sum(hb.values())} -> {sum(ha.values())}")
        for op in sorted(set(hb) | set(ha), key=lambda o: -abs(ha[o] - hb[o])):
            if ha[op] != hb[op]:
                print(f"  {op:34s} {hb[op]:6d} -> {ha[op]:6d} ({ha[op] - hb[op]:+d})")


if __name__ == "__main__":
    main()
