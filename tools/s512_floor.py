#!/usr/bin/env python3
"""Where a 38-us launch goes (VERDICT r05 task 2): C2's shortest shape, S = 512 (B = 16, H = 16, bf16: 512 items, two per
workgroup), back-to-back launches on one stream.

    python tools/s512_floor.py [--seq 512] [--batch 16] [--launches 400] [--out DIR]

Runs itself under `rocprofv3 --kernel-trace` (no counters) and reads the dispatch records: per launch the kernel's own
DURATION (first wave in to last wave out, as the profiler stamps it) and the GAP to the next dispatch of the same stream
(the command processor's barrier between two dependent dispatches, the cache write-back / invalidate at the kernel
boundary, the next dispatch's ramp).  Beside it the event-timed launch INTERVAL of the same loop without the profiler
(what bench.py's C2 number is made of) and -- from lib/trace64_items, if it is there -- the cycles a workgroup spends
between its entry and its exit stamp with the clock they ran at.  interval = duration + gap; duration - the workgroups'
walk = dispatch ramp + drain."""
import argparse
import csv
import glob
import json
import os
import statistics
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    import torch
    import flash_attention
    from flash_helpers import kernel_configs as kc
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(1)
    q, k, v = (torch.randn((args.batch, args.seq, 16, 128), dtype=torch.bfloat16, device=dev, generator=gen) for _ in range(3))
    o = torch.empty_like(q)
    cfg = kc.best_config(kc.DType.BF16, args.seq)
    for _ in range(300):
        flash_attention.forward(cfg, q, k, v, o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = []
    for _ in range(5):
        e0.record()
        for _ in range(args.launches):
            flash_attention.forward(cfg, q, k, v, o)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / args.launches)
    # what ANY kernel boundary costs on this stack: the device-side interval of back-to-back launches of a 64-element kernel,
    # replayed from a graph so that the host is out of the picture (the command processor's dispatch + end-of-kernel release)
    null_us = None
    try:
        x = torch.zeros(64, device=dev)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(3):
                x.add_(1.0)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(200):
                    x.add_(1.0)
            g.replay()
            side.synchronize()
            e0.record(side)
            for _ in range(10):
                g.replay()
            e1.record(side)
            side.synchronize()
        null_us = e0.elapsed_time(e1) * 1e3 / 2000
    except Exception as exc:  # noqa: BLE001
        null_us = "failed: %s" % (str(exc)[:120],)
    flop = 4.0 * args.batch * 16 * args.seq * args.seq * 128
    print(json.dumps({"null_kernel_interval_us": null_us, "interval_us_per_launch": best, "tflops_at_median": flop / (statistics.median(best) * 1e-6) / 1e12,
                      "kernel": str(cfg), "items": args.batch * 16 * (args.seq // 256)}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--launches", type=int, default=400)
    ap.add_argument("--out", default=None)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a)
    me = [sys.executable, os.path.abspath(__file__), "--child", "--seq", str(a.seq), "--batch", str(a.batch), "--launches", str(a.launches)]
    plain = subprocess.run(me, capture_output=True, text=True, timeout=900)
    line = [ln for ln in plain.stdout.splitlines() if ln.startswith("{")]
    if not line:
        print(plain.stdout[-2000:], plain.stderr[-2000:])
        return 1
    ev = json.loads(line[-1])
    tmp = a.out or tempfile.mkdtemp(prefix="s512_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    prof = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, "kt"), "-o", "s512", "--"] + me,
                          capture_output=True, text=True, timeout=1800, env=env, cwd="/tmp")
    files = glob.glob(os.path.join(tmp, "kt", "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print("no kernel trace:", prof.stdout[-1500:], prof.stderr[-1500:])
        return 1
    rows = []
    with open(files[0], newline="") as f:
        for r in csv.DictReader(f):
            if "fa_fwd_kernel64" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort()
    rows = rows[300:]   # (the warm-up launches)
    dur = [(e - s) / 1e3 for s, e in rows]
    gap = [(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(len(rows) - 1)]
    gap = [g for g in gap if g < 50.0]   # (the five timing regions are separated by a synchronize)
    under = [ln for ln in prof.stdout.splitlines() if ln.startswith("{")]
    ev_prof = json.loads(under[-1]) if under else {}
    q = lambda xs, p: sorted(xs)[int(p * (len(xs) - 1))]   # noqa: E731
    print("S = %d, B = %d, H = 16, bf16: %d items on 256 workgroups; kernel %s" % (a.seq, a.batch, ev["items"], ev["kernel"]))
    print("event-timed interval between launches, no profiler (five regions of %d launches): %s us  -> %.1f TFLOP/s at the median"
          % (a.launches, " ".join("%.2f" % x for x in ev["interval_us_per_launch"]), ev["tflops_at_median"]))
    print("back-to-back launches of a 64-element kernel replayed from a graph (what a kernel boundary costs by itself): %s us per launch"
          % (("%.2f" % ev["null_kernel_interval_us"]) if isinstance(ev.get("null_kernel_interval_us"), float) else ev.get("null_kernel_interval_us")))
    if ev_prof:
        print("  ... the same loop under rocprofv3 --kernel-trace: %s us" % " ".join("%.2f" % x for x in ev_prof["interval_us_per_launch"]))
    print("rocprofv3 dispatch records (%d launches): kernel duration median %.2f us (p10 %.2f, p90 %.2f); gap to the next dispatch median %.2f us (p10 %.2f, p90 %.2f)"
          % (len(dur), statistics.median(dur), q(dur, 0.1), q(dur, 0.9), statistics.median(gap), q(gap, 0.1), q(gap, 0.9)))
    print("  duration + gap = %.2f us (the profiled loop's interval)" % (statistics.median(dur) + statistics.median(gap)))
    t64 = os.path.join(ROOT, "flash_attention_from_scratch_amd", "lib", "trace64_items")
    if os.path.exists(t64):
        out = subprocess.run([t64, str(a.seq), str(a.batch), "16"], capture_output=True, text=True, timeout=300).stdout
        for ln in out.splitlines():
            if ln.startswith("==") or ln.startswith("mean over") or ln.startswith("realtime:"):
                print("trace64_items: " + ln[:400])
    return 0


if __name__ == "__main__":
    sys.exit(main())
