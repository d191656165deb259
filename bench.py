#!/usr/bin/env python3
"""bench.py -- the driver's benchmark contract for the FA2-forward hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (flash_attention.forward -> libfa_hip.so) over
one batch of synthetic input already resident in HBM.  At N=1 the workload is
BASELINE.json configs[1] (the headline): bf16, batch 4, heads 16, seq_len 4096,
d_head 128, non-causal.  For N>1 every rank runs that same per-GPU workload on its
own batch shard (global batch = 4*N, no data-path collective, no RCCL traffic in
the timed region): "scaling": "weak".

One JSON line on rank 0:
  value      whole-job TFLOP/s = N * steps * 4*B*H*S^2*d / max-over-ranks time
  roofline   MFMA-bound: achieved TFLOP/s of the kernel from HIP events recorded on
             the launch stream around the K timed launches; peak = 2500 TFLOP/s
             (MI355X dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md)
  cpu_baseline  torch CPU scaled_dot_product_attention (oracle.fa_oracle.sdpa_cpu)
             on the SAME workload, on this host's cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0}  # dense MFMA, MI355X_MICROARCH.md
WORKLOADS = {
    # name: (dtype, per-GPU batch, heads, seq_len, d_head)   -- BASELINE.json configs
    "c1": ("bf16", 4, 16, 4096, 128),      # headline
    "c3": ("fp16", 2, 32, 16384, 128),     # long context
    "c4": ("bf16", 8, 32, 8192, 128),      # the 8-GPU shard (64/8 per GPU)
}
# BASELINE.json configs[2]: bf16 sweep, batch per seq_len from the reference's table
# (py/flash_helpers/test/utils.py:9-16), heads 16, harmonic mean of TFLOP/s
C2_SWEEP = [(512, 16), (1024, 16), (2048, 16), (4096, 16), (8192, 8), (16384, 4)]


def run_c2_sweep(args, device):
    """--workload c2: one JSON line whose value is the harmonic mean over the sweep."""
    import statistics

    import flash_attention
    from flash_helpers import kernel_configs as kc

    per_s = {}
    for seq, batch in C2_SWEEP:
        cfg = kc.parse_kernel_name_into_config(args.kernel) if args.kernel else kc.best_config(kc.DType.BF16, seq)
        gen = torch.Generator(device=device).manual_seed(seq)
        q, k, v = (torch.randn((batch, seq, 16, 128), dtype=torch.bfloat16, device=device, generator=gen)
                   for _ in range(3))
        o = torch.empty_like(q)
        for _ in range(args.warmup):
            flash_attention.forward(cfg, q, k, v, o)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            flash_attention.forward(cfg, q, k, v, o)
        torch.cuda.synchronize(device)
        sec = (time.perf_counter() - t0) / args.steps
        per_s[seq] = {"tflops": mfma_flop(batch, 16, seq, 128) / sec / 1e12, "ms": sec * 1e3,
                      "batch": batch, "kernel": cfg.short_form()}
    value = statistics.harmonic_mean([r["tflops"] for r in per_s.values()])
    print(json.dumps({
        "metric": "bf16 TFLOPs, harmonic mean over seq_len {512..16384}, d_head=128", "value": value,
        "unit": "TFLOP/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": sum(r["ms"] for r in per_s.values()), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "c2: FA2 forward bf16 sweep S in {512,1024,2048,4096,8192,16384}, heads=16, "
                               "batch {16,16,16,16,8,4}, a step = one pass over all six shapes"},
        "per_seq_len": per_s,
        "roofline": {"bound": "mfma", "achieved": value, "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
                     "frac": value / PEAK_TFLOPS["bf16"], "traffic": None},
    }), flush=True)


def mfma_flop(batch, heads, seq, d):
    """Algorithmic FLOPs 4*B*H*S^2*d -- QK^T and PV, non-causal."""
    return 4 * batch * heads * seq * seq * d


def shard_for_rank(global_batch, world, rank):
    """Contiguous batch shard [lo, hi) of rank `rank` (plain batch split)."""
    base, extra = divmod(global_batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def timed_steps(step, steps, warmup, sync, barrier):
    """W untimed steps, then exactly K steps bracketed by barrier + sync. Seconds."""
    for _ in range(warmup):
        step()
    sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    barrier()
    return time.perf_counter() - t0


def max_over_ranks(seconds, world, device):
    if world == 1:
        return seconds
    import torch.distributed as dist

    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def cpu_baseline(dtype, batch, heads, seq, d, budget_s=12.0):
    """torch CPU SDPA on the same workload shape, bounded to ~budget_s."""
    from oracle import fa_oracle as fo  # checker / baseline only -- never the product path

    gen = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn((batch, seq, heads, d), generator=gen).to(dtype) for _ in range(3))
    fo.sdpa_cpu(q, k, v)  # warm-up
    reps, t_total = 0, 0.0
    while reps < 8 and (t_total < budget_s or reps < 2):
        t0 = time.perf_counter()
        fo.sdpa_cpu(q, k, v)
        t_total += time.perf_counter() - t0
        reps += 1
    sec = t_total / reps
    # C0, the reference's own CPU-runnable plumbing case (BASELINE.json configs[0]): fp32 B=2 H=8 S=512
    q0, k0, v0 = (torch.randn((2, 512, 8, 128), generator=gen) for _ in range(3))
    c0 = []
    for _ in range(23):
        t0 = time.perf_counter()
        fo.sdpa_cpu(q0, k0, v0)
        c0.append(time.perf_counter() - t0)
    c0 = sorted(c0[3:])
    return {
        "value": mfma_flop(batch, heads, seq, d) / sec / 1e12,
        "c0_fp32_tflops_median": mfma_flop(2, 8, 512, 128) / c0[len(c0) // 2] / 1e12,
        "c0_fp32_ms_median": c0[len(c0) // 2] * 1e3,
        "unit": "TFLOP/s",
        "cores": torch.get_num_threads(),
        "host_cpus": os.cpu_count(),
        "kind": "port",
        "sample": f"torch CPU SDPA, full workload B={batch} H={heads} S={seq} d={d} "
                  f"{str(dtype).split('.')[-1]}, mean of {reps} reps ({sec * 1e3:.1f} ms each)",
    }


def measured_traffic(kernel_short_form, workload):
    """HBM bytes per launch from the committed rocprofv3 PMC pass of this kernel on this
    workload (profiles/traffic_c1.json, written by tools/gpu_pmc.sh); None if absent."""
    path = os.path.join(ROOT, "profiles", f"traffic_{workload}.json")
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    return rec.get("hbm_bytes_per_launch") if rec.get("kernel") == kernel_short_form else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c1", choices=sorted(WORKLOADS) + ["c2"])
    ap.add_argument("--kernel", default="", help="short-form config; default = best_config(dtype)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) for real multi-GPU runs; gloo lets a 1-GPU box exercise the "
                         "N>1 code path with every rank on cuda:0 (timings then mean nothing)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")

    import flash_attention
    from flash_helpers import kernel_configs as kc

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if args.dist_backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    reduce_device = device if args.dist_backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # RCCL; barrier + max only
        else:
            dist.init_process_group("gloo")

    if args.workload == "c2":
        if world != 1:
            raise SystemExit("the c2 sweep is a single-GPU workload")
        return run_c2_sweep(args, device)
    dtype_name, batch, heads, seq, d = WORKLOADS[args.workload]
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[dtype_name]
    cfg = (kc.parse_kernel_name_into_config(args.kernel) if args.kernel
           else kc.best_config(kc.DType.BF16 if dtype_name == "bf16" else kc.DType.FP16, seq))

    # this rank's shard of the global batch (weak scaling: `batch` per GPU)
    lo, hi = shard_for_rank(batch * world, world, rank)
    gen = torch.Generator(device=device).manual_seed(1000 + rank)
    slab = torch.empty((4, hi - lo, seq, heads, d), dtype=dtype, device=device)
    q, o, k, v = slab[0], slab[1], slab[2], slab[3]   # generate_qkvo layout (utils.py:124-134)
    for t in (q, k, v):
        t.normal_(generator=gen)

    stream = torch.cuda.current_stream(device)

    def step():
        flash_attention.forward(cfg, q, k, v, o)

    def sync():
        torch.cuda.synchronize(device)

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()

    # events on the launch stream (flash_attention launches on torch's current stream)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    state = {"n": 0}

    def step_with_events():
        if state["n"] == 0:
            ev0.record(stream)
        step()
        state["n"] += 1
        if state["n"] == args.steps:
            ev1.record(stream)

    for _ in range(args.warmup):
        step()
    sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_with_events()
    sync()
    barrier()
    seconds = time.perf_counter() - t0
    seconds = max_over_ranks(seconds, world, reduce_device)
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # avg launch duration, this rank

    flop_per_step_rank = mfma_flop(hi - lo, heads, seq, d)
    total_flop = flop_per_step_rank * world * args.steps  # equal shards
    value = total_flop / seconds / 1e12
    achieved = flop_per_step_rank / (kernel_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[dtype_name]

    if rank == 0:
        props = torch.cuda.get_device_properties(device)
        line = {
            "metric": "achieved bf16 TFLOPs and % of MFMA peak at seq_len=4096 d_head=128"
                      if args.workload == "c1" else f"achieved {dtype_name} TFLOPs ({args.workload})",
            "value": value,
            "unit": "TFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": seconds / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype_name,
            "data": "synthetic",
            "pct_of_mfma_peak": 100.0 * value / (peak * world),
            "ref_convention_tflops": value * (4 * d + 6) / (4 * d),  # B*H*(4S^2d+6S^2), kernel_configs.py:102
            "config": {
                "workload": f"{args.workload}: FA2 forward {dtype_name} batch={batch}/GPU heads={heads} "
                            f"seq_len={seq} d_head={d} non-causal",
                "global_batch": batch * world,
                "kernel": cfg.short_form(),
                "parallelism": f"batch-shard x{world}, no collective",
                "device": getattr(props, "gcnArchName", props.name),
                "compute_units": props.multi_processor_count,
            },
            "roofline": {
                "bound": "mfma",
                "achieved": achieved,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": achieved / peak,
                "traffic": measured_traffic(cfg.short_form(), args.workload),
                "algorithmic_bytes": 4 * (hi - lo) * seq * heads * d * 2,
                "kernel_ms": kernel_ms,
                "flop_per_launch": flop_per_step_rank,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(dtype, batch, heads, seq, d)
        print(json.dumps(line), flush=True)

    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
