"""CPU tier: the N>1 path of bench.py (batch shard, barrier-bracketed timing,
max-over-ranks) under torch.distributed with the gloo backend, world_size 2.
The data path has no collective; the step here is the CPU oracle
standing in for the kernel, so that shard boundaries and result assembly are checked
end to end without a GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bench
from oracle import fa_oracle as fo


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, global_batch, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    gen = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn((global_batch, 128, 2, 128), generator=gen).to(torch.bfloat16) for _ in range(3))
    lo, hi = bench.shard_for_rank(global_batch, world, rank)
    qs, ks, vs = (t[lo:hi].contiguous() for t in (q, k, v))
    out = {}

    def step():
        out["o"] = fo.blockwise_forward(qs, ks, vs, 64, 64, n_threads=1)

    seconds = bench.timed_steps(step, steps=2, warmup=1, sync=lambda: None, barrier=dist.barrier)
    slowest = bench.max_over_ranks(seconds + rank, world, torch.device("cpu"))
    torch.save({"lo": lo, "hi": hi, "o": out["o"], "seconds": seconds, "slowest": slowest},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("global_batch", [4, 5])
def test_batch_shard_world_size_2(tmp_path, global_batch):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), global_batch, str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    # shards tile the batch exactly once, in order
    assert parts[0]["lo"] == 0 and parts[-1]["hi"] == global_batch
    assert parts[0]["hi"] == parts[1]["lo"]
    # assembled output == unsharded computation, bit for bit (no cross-batch coupling)
    gen = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn((global_batch, 128, 2, 128), generator=gen).to(torch.bfloat16) for _ in range(3))
    whole = fo.blockwise_forward(q, k, v, 64, 64, n_threads=1)
    assert torch.equal(torch.cat([p["o"] for p in parts]), whole)
    # max-over-ranks: both ranks agree on the slowest rank's time (rank 1 added 1 s)
    assert parts[0]["slowest"] == parts[1]["slowest"] >= parts[1]["seconds"] + 1 - 1e-9


def test_shard_partition_properties():
    for gb in range(1, 70):
        for world in (1, 2, 3, 4, 8):
            spans = [bench.shard_for_rank(gb, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert bench.mfma_flop(4, 16, 4096, 128) == 549755813888


@pytest.mark.parametrize("n", [1, 2])
def test_bench_launches_its_own_ranks(n):
    """`python bench.py --gpus N` with no RANK in the environment starts one rank per GPU itself
    (rendezvous on 127.0.0.1, gloo for the barrier and the max over ranks) and rank 0 prints ONE JSON
    line.  --cpu-dry-run replaces the kernel by a sleep so that the launcher, the shards, the
    barrier-bracketed timing and the per-rank gather run on a CPU-only box."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "5",
                          "--warmup", "2", "--cpu-dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), res.stdout[:500]   # ONE JSON line and nothing else (the gloo banner goes to stderr)
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["steps"] == 5 and rec["warmup"] == 2 and rec["scaling"] == "weak"
    assert len(rec["per_gpu_tflops"]) == n
    assert rec["config"]["global_batch"] == 4 * n
    assert rec["config"]["shards"] == [[4 * r, 4 * r + 4] for r in range(n)]
    assert 2.0 <= rec["ms_per_step"] < 50.0          # five 2-ms sleeps per rank, max over ranks
    # the aggregate is N ranks' work over the slowest rank's time: never more than the sum of the ranks
    assert rec["value"] <= sum(rec["per_gpu_tflops"]) * (1 + 1e-9)


def test_scale_command_dry_run_eight_ranks_on_c4():
    """The documented SCALE command is `python bench.py --gpus 8 --workload c4` (BASELINE.json configs[4]: B=64 H=32
    S=8192 bf16 as eight batch shards of 8, no collective).  Its launcher, shard arithmetic, barrier-bracketed timing and
    per-rank gather run here with eight gloo ranks and a sleeping step (--cpu-dry-run); no --steps: the default applies."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--workload", "c4", "--warmup", "1",
                          "--cpu-dry-run", "--rocprof-rank", "3"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["scaling"] == "weak" and len(rec["per_gpu_tflops"]) == 8
    assert rec["config"]["global_batch"] == 64 and rec["config"]["heads"] == 32 and rec["config"]["seq_len"] == 8192
    assert rec["config"]["shards"] == [[8 * r, 8 * r + 8] for r in range(8)]
    assert rec["config"]["flop_per_step_per_gpu"] == 8796093022208          # C4 per GPU
    assert 8 * rec["config"]["flop_per_step_per_gpu"] == 70368744177664     # ... and the whole job
    assert rec["value"] <= sum(rec["per_gpu_tflops"]) * (1 + 1e-9)
    # round 5: every rank's launcher is pinned to its own CPUs (on the GPU box: a slice of the NUMA node of
    # its GPU; here: of whatever this box allows), and the line says where
    aff = [g["affinity"] for g in rec["per_gpu"]]
    assert len(aff) == 8 and all(a and a["pinned"] and a["n_cpus"] >= 1 for a in aff), aff
    if len(os.sched_getaffinity(0)) >= 8:
        assert len({a["cpus"] for a in aff}) == 8, aff          # distinct, non-overlapping
    assert [a["rank"] for a in aff] == list(range(8))
    # --rocprof-rank 3: the command rank 3 would replace itself with on the GPU box (not executed in a dry run)
    cmd = rec["rocprof_command"]
    assert rec["rocprof_rank"] == 3 and cmd[:3] == ["rocprofv3", "--kernel-trace", "--stats"] and "scale_rank3" in cmd
    assert "--pmc" not in cmd and cmd[cmd.index("--") + 3:][:4] == ["--gpus", "8", "--workload", "c4"]
