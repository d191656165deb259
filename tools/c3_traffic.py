#!/usr/bin/env python3
"""Where do the 1.5x algorithmic bytes of C3 (fp16 B=2 H=32 S=16384) come from, and do they reach HBM?
One rocprofv3 PMC pass per counter group over `bench.py --traffic-child --workload <w>` (counters only), for C3 and, as
the control, C1: read requests of the L2 towards the fabric by size (32 / 64 / 128 B) and by destination (DRAM / GMI / IO),
L2 hits and misses, write requests.  rocprofv3 -L on this box lists no Infinity-Cache (MALL) or memory-controller (UMC)
counter, so a MALL hit and an HBM access are both "a read request destined for DRAM" here; what the counters can show is
that the excess is K / V re-read by the second round of a head's Q blocks (the request count is 1.5x, all of it
DRAM-destined 128-B requests), and the reuse distance says where it is served from.
Usage: python tools/c3_traffic.py > profiles/r03/c3_traffic.txt"""
import sys

sys.path.insert(0, ".")
import bench  # noqa: E402

GROUPS = [["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_DRAM_sum"],
          ["TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_BUBBLE_sum"],
          ["TCC_EA0_RDREQ_GMI_32B_sum", "TCC_EA0_RDREQ_IO_32B_sum", "TCC_EA0_RDREQ_DRAM_32B_sum"],
          ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"],
          ["FETCH_SIZE"], ["WRITE_SIZE"]]


def main():
    for workload in ("c3", "c1"):
        dtype_name, batch, heads, seq, d = bench.WORKLOADS[workload]
        alg = 4 * batch * seq * heads * d * 2
        print(f"== {workload}: {dtype_name} B={batch} H={heads} S={seq}: algorithmic bytes {alg} (Q + K + V read once, O written once)")
        got = {}
        for group in GROUPS:
            res, why = bench.rocprof_pass(group, ["--workload", workload])
            if res is None:
                print("   ", " ".join(group), "->", why)
                continue
            got.update(res)
            print("   ", "  ".join(f"{c} {res[c]:.0f}" for c in group), f" (kernel {res['duration_ns'] * 1e-3:.1f} us)")
        if "TCC_EA0_RDREQ_128B_sum" in got and "TCC_EA0_RDREQ_64B_sum" in got:
            by_size = 128 * got["TCC_EA0_RDREQ_128B_sum"] + 64 * got["TCC_EA0_RDREQ_64B_sum"] + 32 * got.get("TCC_EA0_RDREQ_32B_sum", 0)
            print(f"    read bytes by request size (128 / 64 / 32 B): {by_size:.0f} = {by_size / (0.75 * alg):.3f} x the algorithmic READ bytes (3/4 of the total)")
        if "FETCH_SIZE" in got and "WRITE_SIZE" in got:
            tot = (2 * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024
            print(f"    2 x FETCH_SIZE + WRITE_SIZE = {tot:.0f} = {tot / alg:.3f} x algorithmic (bench.py's roofline.traffic)")
        if "TCC_HIT_sum" in got:
            print(f"    L2 hit rate {got['TCC_HIT_sum'] / (got['TCC_HIT_sum'] + got['TCC_MISS_sum']):.3f}")


if __name__ == "__main__":
    main()
