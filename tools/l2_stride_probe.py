#!/usr/bin/env python3
"""Why does the alternating K / V walk (ALT) find so little of a head's tail in the L2?  (VERDICT r05 task 5.)

S = 16384, fp16, batch x heads = 64 every time -- the same items, the same walk -- but the heads dimension, i.e. the STRIDE
between two consecutive K / V rows of one head in the (batch, seq, heads, 128) layout, varies: heads = 1 (rows contiguous,
256 B apart), 8 (2 KiB), 32 (8 KiB = C3).  For each: FETCH_SIZE of one launch (rocprofv3 --pmc, counters only) through the
alternating form and, with the launcher's measurement switch FA_HIP_NO_ALT, through the plain form.  If what the second round of
a head still finds in the L2 grows as the stride shrinks, the 4 MiB are there but a strided head only reaches a slice of the
sets; if not, it is the replacement.          python tools/l2_stride_probe.py > profiles/r06/l2_stride_probe.txt"""
import csv
import glob
import os
import shutil
import statistics
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S = 16384


def child(batch, heads):
    import torch
    import flash_attention
    from flash_helpers import kernel_configs as kc
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(3)
    q, k, v = (torch.randn((batch, S, heads, 128), dtype=torch.float16, device=dev, generator=gen) for _ in range(3))
    o = torch.empty_like(q)
    cfg = kc.best_config(kc.DType.FP16, S)
    for _ in range(3):
        flash_attention.forward(cfg, q, k, v, o)
    torch.cuda.synchronize()


def one_pass(counter, batch, heads, no_alt):
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    out_dir = tempfile.mkdtemp(prefix="l2p_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("FA_HIP_NO_ALT", None)
    if no_alt:
        env["FA_HIP_NO_ALT"] = "1"
    cmd = [prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out_dir, "-o", "p", "--",
           sys.executable, os.path.abspath(__file__), "--child", str(batch), str(heads)]
    subprocess.run(cmd, cwd="/tmp", env=env, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    vals, names, dur = [], set(), []
    for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if "fa_fwd" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    vals.append(float(row["Counter_Value"]))
                    names.add(row["Kernel_Name"])
                    dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    shutil.rmtree(out_dir, ignore_errors=True)
    return statistics.mean(vals), statistics.mean(dur) * 1e-3, sorted(names)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]), int(sys.argv[3]))
    print("S = 16384 fp16, batch x heads = 64 (4096 items of 256 visits; a head = 64 Q blocks = two rounds of an XCD's 32 workgroups; K + V of a head 8 MiB, L2 4 MiB)")
    print("read factor = 2 x FETCH_SIZE KiB / (Q + K + V bytes); 1.00 = everything read once, 1.667 = K and V read twice")
    for batch, heads in ((64, 1), (8, 8), (2, 32)):
        alg_read = 3 * batch * S * heads * 128 * 2
        for no_alt in (True, False):
            fetch, us, names = one_pass("FETCH_SIZE", batch, heads, no_alt)
            form = "plain walk      " if no_alt else "alternating walk"
            alt_ran = any(n.rstrip(")").endswith("true>(fa::KernelArgs") or ", true>" in n for n in names)
            print("heads %2d (row stride %5d B)  %s  read factor %.3f  K / V re-read %.3f x   kernel %.0f us   [%s]"
                  % (heads, heads * 256, form, 2 * fetch * 1024 / alg_read, (2 * fetch * 1024 / alg_read - 1 / 3) / (2 / 3),
                     us, "ALT form" if alt_ran else "plain form"))


if __name__ == "__main__":
    sys.exit(main())
