"""isa_lint64.py -- hazard lint for the hand-placed 64-rows-per-wave kernel.

hipcc treats the inline-asm MFMAs as opaque single-cycle instructions, so it neither pads the
hazards around them nor keeps their operand registers allocated while the matrix pipe is still
reading them.  This script disassembles nothing: it reads the `-S` output and reports

  WAR   an instruction within `--window` instructions AFTER an MFMA writes a VGPR that the MFMA
        reads as A or B operand (seen on hardware as rare one-ulp run-to-run differences);
  RAW   an MFMA reads as A / B a VGPR written by a VALU instruction fewer than `--raw` instructions
        earlier (v_cvt_pk -> MFMA needs wait states hipcc does not insert for asm);
  AGPR  compiler-generated v_accvgpr_* inside the main loop (accumulators must stay put);
  MFMAD an instruction that reads or writes a register of an MFMA's result fewer than passes + 3 wait states behind
        it (11 for the 8-pass 32x32x16, 7 for 16x16x32): the result is not there yet, and nothing interlocks.  An MFMA
        that accumulates into exactly the same registers (its C operand) is the one legal back-to-back user;
  M0GAP an LDS-DMA (global_load_lds / buffer_load ... lds) or other M0 reader issued right behind the scalar write of M0,
        with no instruction in between (one wait state is required; the pieces' M0 writes sit one gap early by plan);
  PERMSW a v_permlane16/32_swap fewer than 2 wait states behind a vector instruction that wrote one of its two
        registers (hipcc pads this when it knows the producer; the row-max / row-sum producers are asm);
  STDATA a vector instruction that writes a data register of a global / flat / buffer store of more than 64 bits
        within 2 wait states behind it (the store still reads them; hipcc pads this only for its own stores, the
        epilogue's are asm: seen as garbage rows in O when hipcc reused v[i] for the next store's address);
  SGPRVM a vector-memory instruction that reads, as its scalar base, an SGPR written by a vector instruction
        (v_readlane / v_readfirstlane: hipcc's reloads of scalars it parked in VGPR lanes) fewer than 5 wait states
        earlier -- hipcc pads this for its own memory instructions, not for the DMA pieces and stores in asm;
  M0    an instruction hipcc itself emitted (outside ;;#ASMSTART / ;;#ASMEND) that reads or writes M0 in
        a fa_fwd_kernel64 function: the DMA pieces leave their LDS destination in M0 across statements.

Usage: python isa_lint64.py kernel.s [--window 3] [--raw 2] [--only fa_fwd_kernel64]
The library build runs it over the 64-rows-per-wave slices (csrc/Makefile) and fails on any finding.
"""
import argparse
import re
import sys


def regs(tok):
    tok = tok.strip(",")
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def regs2(tok):
    """-> set of ('v' | 'a', index) named by one operand token (v7, v[4:7], a3, a[0:15]); empty for anything else."""
    tok = tok.strip(",")
    m = re.match(r"([va])\[(\d+):(\d+)\]$", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([va])(\d+)$", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


M0_USERS = re.compile(r"\bm0\b|^s_movrel|^v_movrel|^s_set_gpr_idx")


def split_kernels(path):
    """-> [(function name, [instruction], [index of the instructions hipcc itself emitted])]: one entry per
    function of the `-S` output; instructions between ;;#ASMSTART and ;;#ASMEND are the source's own."""
    kernels, cur, own, name, in_asm = [], [], [], "", False
    for raw_line in open(path).read().split("\n"):
        l = raw_line.strip()
        if l.startswith(";;#ASMSTART"):
            in_asm = True
        elif l.startswith(";;#ASMEND"):
            in_asm = False
        m = re.match(r"^(_Z\w+):", l)
        if m and not cur:
            name = m.group(1)
        if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
            if l.startswith(".Lfunc_end"):
                kernels.append((name, cur, own))
                cur, own, name = [], [], ""
            continue
        if not in_asm:
            own.append(len(cur))
        cur.append(l)
    if cur:
        kernels.append((name, cur, own))
    return kernels


def lint(path, window=3, raw=2, only=None):
    """Findings [(kind, kernel index, instruction index, mfma / context, offender)].  only: restrict to
    functions whose (mangled) name contains this string."""
    findings = []
    kernels = []
    names = []
    for name, code, own in split_kernels(path):
        if only and only not in name:
            continue
        kidx = len(kernels)
        kernels.append(code)
        names.append(name)
        # M0: the hand-placed DMA writes M0 (the LDS destination) in one asm statement and reads it in a
        # later one, with no save / restore.  That is only sound while hipcc itself never touches M0 in
        # the function (it is compiler-reserved and NOT preserved around asm statements): no dynamic
        # register indexing (s_set_gpr_idx, movrel), no sendmsg / LDS-DMA builtin that needs it.
        if "fa_fwd_kernel64" in name or not name:
            for i in own:
                if M0_USERS.search(code[i]):
                    findings.append(("M0", kidx, i, name, code[i]))
    for kidx, code in enumerate(kernels):
        for i, l in enumerate(code):
            if not l.startswith("v_mfma"):
                continue
            ops = l.split()[1:]
            rd = regs(ops[1]) | regs(ops[2])
            slots, k = 0, 0  # issue slots behind the MFMA: an `s_nop n` fills n + 1 of them
            while slots < window:
                k += 1
                if i + k >= len(code) or code[i + k].startswith("v_mfma"):
                    break
                n = code[i + k]
                slots += 1 + (int(n.split()[1]) if n.startswith("s_nop") else 0)
                if slots > window:
                    break
                if n.startswith("v_") or n.startswith("ds_read") or n.startswith("global_load") or n.startswith("scratch_load"):
                    written = regs(n.split()[1])
                    if n.startswith("v_permlane32_swap") or n.startswith("v_swap"):
                        written |= regs(n.split()[2])  # these exchange: both operands are written
                    if written & rd:
                        findings.append(("WAR", kidx, i, l, n))
            dist = 0  # issue slots between producer and MFMA: an `s_nop n` fills n + 1 of them
            k = 1
            while i - k >= 0 and not code[i - k].startswith("v_mfma"):
                p = code[i - k]
                dist += 1
                if dist > raw:
                    break
                if p.startswith("s_nop"):
                    dist += int(p.split()[1])
                elif p.startswith("v_") and not p.startswith("v_cmp"):
                    if regs(p.split()[1]) & rd:
                        findings.append(("RAW", kidx, i, l, p))
                k += 1
        # MFMAD: MFMA result -> any other use, passes + 3 wait states
        for i, l in enumerate(code):
            if not l.startswith("v_mfma"):
                continue
            ops = [o.strip(",") for o in l.split()[1:5]]
            dst = regs2(ops[0])
            need = 7 if "16x16x32" in l else 11
            slots, k = 0, 0
            while slots < need and i + k + 1 < len(code):
                k += 1
                n = code[i + k]
                if n.startswith(("s_branch", "s_endpgm", "s_setpc")):
                    break  # what follows in the listing is another path
                toks = [o.strip(",") for o in n.split()[1:]]
                if n.startswith("v_mfma"):
                    ab = regs2(toks[1]) | regs2(toks[2])
                    c = regs2(toks[3]) if len(toks) > 3 else set()
                    if (ab & dst) or ((c & dst) and c != dst) or ((regs2(toks[0]) & dst) and regs2(toks[0]) != dst):
                        findings.append(("MFMAD", kidx, i, l, n))
                elif not n.startswith("s_") and not n.startswith(";"):
                    used = set()
                    for t in toks:
                        used |= regs2(t)
                    if used & dst:
                        findings.append(("MFMAD", kidx, i, l, n))
                slots += 1 + (int(n.split()[1]) if n.startswith("s_nop") else 0)
        # M0GAP: s_mov m0 -> LDS-DMA needs one wait state
        for i, l in enumerate(code[:-1]):
            if re.match(r"s_(mov|add|or|and|lshl)\w*\s+m0\b", l):
                n = code[i + 1]
                if re.match(r"global_load_lds|buffer_load\w+.*\blds\b|ds_gws|s_sendmsg", n):
                    findings.append(("M0GAP", kidx, i, l, n))
        # PERMSW: VALU result -> v_permlane*_swap of it: 2 wait states
        for i, l in enumerate(code):
            m = re.match(r"v_permlane(?:16|32)_swap\S*\s+(\S+)\s+(\S+)", l)
            if not m:
                continue
            rd = regs(m.group(1)) | regs(m.group(2))
            slots, k = 0, 0
            while slots < 2 and i - k - 1 >= 0:
                k += 1
                p = code[i - k]
                if p.startswith(("s_branch", "s_endpgm", "s_setpc", "s_cbranch")):
                    break
                if p.startswith("v_") and not p.startswith(("v_cmp", "v_readlane", "v_readfirstlane")) and regs(p.split()[1]) & rd:
                    findings.append(("PERMSW", kidx, i, l, p))
                slots += 1 + (int(p.split()[1]) if p.startswith("s_nop") else 0)
        # SGPRVM: VALU write of an SGPR -> vector-memory read of it as the scalar base: 5 wait states
        for i, l in enumerate(code):
            m = re.match(r"(v_readlane_b32|v_readfirstlane_b32)\s+s(\d+)", l)
            if not m:
                continue
            sreg, slots, k = int(m.group(2)), 0, 0
            while slots < 5 and i + k + 1 < len(code):
                k += 1
                n = code[i + k]
                if n.startswith(("s_branch", "s_endpgm", "s_setpc")):
                    break
                w = re.match(r"s_(?!nop|waitcnt|cmp|cbranch|branch|barrier|bitcmp)\w+\s+s(?:(\d+)|\[(\d+):(\d+)\])", n)
                if w and (int(w.group(1)) == sreg if w.group(1) else int(w.group(2)) <= sreg <= int(w.group(3))):
                    break  # rewritten by a scalar instruction: the memory instruction reads that result
                if re.match(r"(global|flat|scratch|buffer)_", n):
                    if any(int(a) <= sreg <= int(b) for a, b in re.findall(r"s\[(\d+):(\d+)\]", n)):
                        findings.append(("SGPRVM", kidx, i, n, l))
                        break
                slots += 1 + (int(n.split()[1]) if n.startswith("s_nop") else 0)
        # STDATA: wide stores and the two wait states behind them
        for i, l in enumerate(code):
            m = re.match(r"(global|flat|scratch)_store_dwordx[34]\s+\S+\s+(\S+)|buffer_store_dwordx[34]\s+(\S+)", l)
            if not m:
                continue
            data = regs(m.group(2) or m.group(3))
            slots, k = 0, 0
            while slots < 2 and i + k + 1 < len(code):
                k += 1
                n = code[i + k]
                if n.startswith(("s_branch", "s_endpgm", "s_setpc")):
                    break
                if n.startswith("v_") and not n.startswith("v_cmp") and not n.startswith("v_readlane") and not n.startswith("v_readfirstlane"):
                    if regs(n.split()[1]) & data:
                        findings.append(("STDATA", kidx, i, l, n))
                slots += 1 + (int(n.split()[1]) if n.startswith("s_nop") else 0)
        # AGPR: a visit is 32 MFMAs into VGPRs (S) followed by 32 into AGPRs (O).  Behind its third MFMA
        # (the rescale of O sits in front of it) no compiler-made accumulator copy may appear: hipcc has
        # been seen hoisting the rescale path's 128 v_accvgpr_read to the end of the PREVIOUS visit,
        # 16 behind every P.V MFMA, each waiting for that MFMA to retire (a trace build; +25 % per visit).
        # The one-Q-tile-per-wave form (template argument QTP = 1, round 5) has visits of 16 + 16.
        mf = [i for i, l in enumerate(code) if l.startswith("v_mfma")]
        kinds = "".join("a" if code[i].split()[1].startswith("a[") else "v" for i in mf)
        # (round 6: the 64-row speculative plain form redoes failed items as half items with one-tile visits: both shapes
        # are looked for in every kernel; a 16 + 16 window that lies inside a 32 + 32 visit is that visit's, not one more)
        covered = []
        for half in (32, 16):
          pos = 0
          while True:
              j = kinds.find("v" * half + "a" * half, pos)
              if j < 0:
                  break
              if half == 32:
                  covered.append((j, j + 64))
              elif any(lo <= j and j + 32 <= hi for lo, hi in covered):
                  pos = j + 1
                  continue
              # (a copy that touches none of the visit's own matrix registers -- O, the accumulators, and Q, the B operands --
              # is not one of those: the pre-scaled Q of the NEXT item is written into the spare Q set on the slow path of an
              # item's first visits, through VGPRs, while the MFMAs work on the current set)
              used = set()
              for i in mf[j:j + 2 * half]:
                  for tok in code[i].split()[1:]:
                      used |= {r for r in regs2(tok) if r[0] == "a"}
              for i in range(mf[j + 2], mf[j + 2 * half - 1] + 1):
                  if code[i].startswith("v_accvgpr_"):
                      touched = set()
                      for tok in code[i].split()[1:]:
                          touched |= {r for r in regs2(tok) if r[0] == "a"}
                      if touched & used:
                          findings.append(("AGPR", kidx, i, code[mf[j + 2 * half - 1]], code[i]))
                          break
              pos = j + 2 * half
    return findings


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--window", type=int, default=3)
    ap.add_argument("--raw", type=int, default=2)
    ap.add_argument("--only", default=None, help="only functions whose mangled name contains this")
    a = ap.parse_args()
    f = lint(a.asm, a.window, a.raw, a.only)
    for kind, k, i, m, o in f:
        print(f"{kind} kernel#{k} @{i}: {m}   <->   {o}")
    print(f"{len(f)} finding(s)")
    sys.exit(1 if f else 0)
