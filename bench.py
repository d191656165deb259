#!/usr/bin/env python3
"""bench.py -- the driver's benchmark contract for the FA2-forward hot path.

    python bench.py --gpus N --steps K --warmup W          (N > 1 spawns one rank per GPU itself)
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
             the launch stream around the K timed launches (and after every launch: the
             per-launch distribution); peak = 2500 TFLOP/s (MI355X dense bf16 MFMA,
             /opt/skills/guides/MI355X_MICROARCH.md); traffic = HBM bytes per launch from
             rocprofv3 PMC passes over this same command (FETCH_SIZE x 2 + WRITE_SIZE);
             pipe_counters = matrix-pipe busy fraction, instructions per MFMA and the
             GRBM_GUI_ACTIVE-derived effective clock from a third pass
  clocks     shader / memory clock and socket power sampled from the GPU's hwmon files
             during the timed region (SURVEY.md 8d asks for them: the chip clocks to its
             power budget)
  cpu_baseline  torch CPU scaled_dot_product_attention (oracle.fa_oracle.sdpa_cpu)
             on the SAME workload, on this host's cores, rank 0, N=1 only.

  roofline.mfma_only_*  the matrix pipe's practical roof on THIS box in THIS run: a register-only MFMA
             loop (tools/mfma_energy.hip `quick`: the kernel's issue order, no memory traffic, ~20 ms
             launches) on zero and on N(0,1) operands, bf16 and fp16 -- the chip clocks to its power
             budget, and random operands cost ~30 % of the datasheet rate before any kernel is involved
  protocols  the same kernel under the reference's own timing protocol as well (pt_bench.py:145-174:
             cache flush + idle spin before every launch, >= 50 reps): `value` is back-to-back launches
             (what the driver's wall clock sees), `protocols.hermetic` the other figure, same run
  speculative  fa_fwd_stats of the timed launches: work items and how many the speculative softmax
             computed twice (--data sink / heavy show the cliff on non-Gaussian logits)

--hermetic makes the hermetic protocol the headline `value` instead.
"""
import argparse
import glob
import json
import os
import socket
import statistics
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0}  # dense MFMA, MI355X_MICROARCH.md
MAX_SCLK_MHZ = 2400.0                            # MI355X_MICROARCH.md, chip-level parameters
WORKLOADS = {
    # name: (dtype, per-GPU batch, heads, seq_len, d_head)   -- BASELINE.json configs
    "c1": ("bf16", 4, 16, 4096, 128),      # headline
    "c3": ("fp16", 2, 32, 16384, 128),     # long context
    "c4": ("bf16", 8, 32, 8192, 128),      # the 8-GPU shard (64/8 per GPU)
}
# BASELINE.json configs[2]: bf16 sweep, batch per seq_len from the reference's table
# (py/flash_helpers/test/utils.py:9-16), heads 16, harmonic mean of TFLOP/s
C2_SWEEP = [(512, 16), (1024, 16), (2048, 16), (4096, 16), (8192, 8), (16384, 4)]


def mfma_flop(batch, heads, seq, d):
    """Algorithmic FLOPs 4*B*H*S^2*d -- QK^T and PV, non-causal."""
    return 4 * batch * heads * seq * seq * d


def make_inputs(data, shape, dtype, device, gen):
    """Synthetic q, k, v of `shape` = (batch, seq, heads, d).  `randn`: N(0, 1) (the reference's benchmark data,
    utils.py:112-121).  `sink`: the same plus an attention sink -- one head dimension carries a constant in every
    query and in the FIRST FOUR keys, worth +12 nats of logit (= 17 binades) over everything else: the shape of
    BOS / sink keys in trained models, and the speculative softmax's worst case (those keys are visited LAST).
    `heavy`: K drawn from a Student-t with 3 degrees of freedom (unit variance): occasional large logits anywhere."""
    q, k, v = (torch.empty(shape, dtype=dtype, device=device) for _ in range(3))
    for t in (q, k, v):
        t.normal_(generator=gen)
    d = shape[-1]
    if data == "sink":
        a = (12.0 * d ** 0.5) ** 0.5           # a * a / sqrt(d) = 12 nats
        q[..., 0] = a
        k[..., 0] = 0
        k[:, :4, :, 0] = a
    elif data == "heavy":
        z = torch.empty(shape, dtype=torch.float32, device=device).normal_(generator=gen)
        chi = sum(torch.empty(shape, dtype=torch.float32, device=device).normal_(generator=gen) ** 2 for _ in range(3))
        k.copy_((z / (chi / 3).sqrt() / 3 ** 0.5).to(dtype))   # t_3 has variance 3
    elif data != "randn":
        raise SystemExit(f"unknown --data {data}")
    return q, k, v


def mfma_only_roof(device_index):
    """Run lib/mfma_energy quick on `device_index` (HIP_VISIBLE_DEVICES narrows the child to it): -> {"bf16_zeros": ...,
    "bf16_normal": ..., "fp16_zeros": ..., "fp16_normal": ...} TFLOP/s, or {"error": ...}."""
    exe = os.path.join(ROOT, "flash_attention_from_scratch_amd", "lib", "mfma_energy")
    if not os.path.exists(exe):
        return {"error": "lib/mfma_energy not built (make -C flash_attention_from_scratch_amd/csrc tools)"}
    env = dict(os.environ)
    vis = env.get("HIP_VISIBLE_DEVICES") or env.get("ROCR_VISIBLE_DEVICES")
    if vis is None:
        env["HIP_VISIBLE_DEVICES"] = str(device_index)
    try:
        out = subprocess.run([exe, "quick"], env=env, capture_output=True, text=True, timeout=120, check=True).stdout
    except (subprocess.SubprocessError, OSError) as exc:
        return {"error": f"mfma_energy quick: {type(exc).__name__}"}
    words = out.replace("mfma_energy quick:", "").split()
    got = {}
    for name, val in zip(words[0::2], words[1::2]):
        try:
            got[name] = float(val)
        except ValueError:
            break
    return got if len(got) == 4 else {"error": "unparsed: " + out.strip()[:200]}


class stdout_to_stderr:
    """Within the block, file descriptor 1 points at stderr: the gloo transport prints its connection banner
    ('[Gloo] Rank 0 is connected to ...') to the C stdout, and rank 0's stdout must carry ONE JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def shard_for_rank(global_batch, world, rank):
    """Contiguous batch shard [lo, hi) of rank `rank` (plain batch split)."""
    base, extra = divmod(global_batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def timed_steps(step, steps, warmup, sync, barrier):
    """W untimed steps, then exactly K steps bracketed by a barrier + synchronize on both sides.  Seconds of THIS rank;
    the caller takes the max over ranks.  The closing pair is barrier THEN synchronize: the launches are asynchronous,
    so the host-side barrier (a gloo round trip is 0.1-1 ms, 10 % of a 20-step region) runs while the devices still
    work, and the clock stops when this rank's device goes idle."""
    for _ in range(warmup):
        step()
    sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    sync()
    return time.perf_counter() - t0


HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak (6.3 TB/s achievable)
SUSTAINED_BELOW_S = 0.1     # a timed region shorter than this also gets the sub-measurement below
SUSTAINED_TARGET_S = 0.25


def sustained_region(step, sync, barrier, seconds, steps, world, device):
    """When the contract's K steps are a region of a few milliseconds (C1: 20 x 0.43 ms), the max-over-ranks wall time is
    mostly the host's -- a gloo barrier alone is 0.1-1 ms -- and a 1 -> 8 GPU curve built on it measures jitter.  So every
    rank then also times a region of >= 250 ms of the same steps, bracketed exactly like the main one, and the line carries
    it as `sustained` (same units; `value` stays the contract's).  `seconds` is already the max over ranks, so every rank
    takes the same branch and the same step count."""
    if seconds >= SUSTAINED_BELOW_S:
        return None
    n2 = max(steps, int(SUSTAINED_TARGET_S / max(seconds / steps, 1e-7)) + 1)
    s2 = max_over_ranks(timed_steps(step, n2, 0, sync, barrier), world, device)
    return {"steps": n2, "seconds": s2, "ms_per_step": s2 / n2 * 1e3,
            "why": f"the {steps}-step region was {seconds * 1e3:.1f} ms (< {SUSTAINED_BELOW_S * 1e3:.0f} ms): host jitter and "
                   "barrier latency are a visible share of it; this region has the same brackets and max-over-ranks"}


def max_over_ranks(seconds, world, device):
    if world == 1:
        return seconds
    import torch.distributed as dist

    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def distribution(samples_ms):
    """mean / median / min / max / stddev of per-launch milliseconds (pt_bench.py:38-79's fields)."""
    return {
        "n": len(samples_ms),
        "mean": statistics.mean(samples_ms),
        "median": statistics.median(samples_ms),
        "min": min(samples_ms),
        "max": max(samples_ms),
        "stddev": statistics.stdev(samples_ms) if len(samples_ms) > 1 else 0.0,
    }


# ---- clocks and power from the GPU's hwmon files --------------------------------------------
def hwmon_dir(device_index):
    """hwmon directory of torch device `device_index` (matched by PCI address), or None."""
    try:
        props = torch.cuda.get_device_properties(device_index)
        want = "%04x:%02x:%02x." % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
    except Exception:
        want = None
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
    picked = None
    for dev in cards:
        real = os.path.realpath(dev)
        if want and want in real:
            picked = dev
            break
    if picked is None and cards and not want:
        picked = cards[min(device_index, len(cards) - 1)]
    if picked is None:
        return None
    hw = sorted(glob.glob(os.path.join(picked, "hwmon", "hwmon*")))
    return hw[0] if hw else None


class ClockSampler:
    """Background thread: shader clock (freq1), memory clock (freq2) and socket power (power1) every
    `period` seconds between start() and stop().  Reading three sysfs files costs the host a few
    microseconds and the GPU nothing."""

    FILES = {"sclk_mhz": ("freq1_input", 1e-6), "mclk_mhz": ("freq2_input", 1e-6), "power_w": ("power1_input", 1e-6)}

    def __init__(self, hw, period=0.002):
        self.hw, self.period = hw, period
        self.samples = {k: [] for k in self.FILES}
        self.stamps = {k: [] for k in self.FILES}   # time.perf_counter() of every sample: summary() can keep a window
        self._stop = threading.Event()
        self._thread = None

    def _read(self):
        for key, (name, scale) in self.FILES.items():
            try:
                with open(os.path.join(self.hw, name)) as f:
                    self.samples[key].append(float(f.read().strip()) * scale)
                    self.stamps[key].append(time.perf_counter())
            except (OSError, ValueError):
                pass

    def _run(self):
        while not self._stop.is_set():
            self._read()
            self._stop.wait(self.period)

    def start(self):
        if self.hw:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def stop(self):
        if self._thread:
            self._stop.set()
            self._thread.join()
            self._read()  # one sample at the end of the region even if it was shorter than a period

    def summary(self, t0=None, t1=None):
        """Mean / min / max per quantity; with t0 (and t1) only the samples taken inside that perf_counter window -- the
        thread may be started before the timed region so that its start-up does not sit between the opening bracket and the
        first launch (the GPU drops its clock within a millisecond of idling)."""
        out = {"source": (self.hw or "unavailable") + " (freq1_input, freq2_input, power1_input)"}
        for key, all_vals in self.samples.items():
            vals = [v for v, ts in zip(all_vals, self.stamps[key]) if (t0 is None or ts >= t0) and (t1 is None or ts <= t1)]
            if not vals and all_vals and t0 is not None:
                vals = all_vals[-1:]   # (a region shorter than one period: the sample taken at its end)
            if vals:
                out[key] = {"mean": statistics.mean(vals), "min": min(vals), "max": max(vals), "n": len(vals)}
        try:
            with open(os.path.join(self.hw, "power1_cap")) as f:
                out["power_cap_w"] = float(f.read().strip()) * 1e-6
        except (OSError, ValueError, TypeError):
            pass
        return out


# ---- HBM traffic of one launch, measured: rocprofv3 PMC passes over this command -------------
def committed_traffic(kernel_short_form, workload):
    """Fallback: the committed PMC result of tools/gpu_pmc.sh (profiles/traffic_<workload>.json)."""
    path = os.path.join(ROOT, "profiles", f"traffic_{workload}.json")
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    return rec.get("hbm_bytes_per_launch") if rec.get("kernel") == kernel_short_form else None


def rocprof_pass(counters, argv_tail, timeout_s=150):
    """One `rocprofv3 --kernel-trace --pmc <counters>` pass (counters only: no other trace domain) over
    `bench.py --traffic-child <same workload>`; -> ({counter: mean over the kernel's dispatches,
    "duration_ns": mean dispatch duration in that pass}, None) or (None, reason)."""
    import csv
    import shutil
    import tempfile

    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    out_dir = tempfile.mkdtemp(prefix="fa_pmc_", dir="/tmp")
    cmd = [prof, "--kernel-trace", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", out_dir, "-o", "p", "--",
                                                                sys.executable, os.path.abspath(__file__), "--traffic-child"] + argv_tail
    try:
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=timeout_s,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    except (subprocess.SubprocessError, OSError) as exc:
        shutil.rmtree(out_dir, ignore_errors=True)
        return None, f"rocprofv3 --pmc {' '.join(counters)}: {type(exc).__name__}"
    vals, durations = {c: [] for c in counters}, {}
    for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if "fa_fwd" in row.get("Kernel_Name", "") and row.get("Counter_Name") in vals:
                    vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
                    durations[row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    shutil.rmtree(out_dir, ignore_errors=True)
    if any(not v for v in vals.values()):
        return None, "no rows for the kernel: " + ", ".join(c for c, v in vals.items() if not v)
    out = {c: statistics.mean(v) for c, v in vals.items()}
    out["duration_ns"] = statistics.mean(durations.values())
    return out, None


def measure_traffic(argv_tail):
    """HBM bytes per launch of the dominant kernel, measured: FETCH_SIZE and WRITE_SIZE each take more than
    half of the TCC counter slots, so two passes; 2 * FETCH_SIZE + WRITE_SIZE, both reported in KiB
    (MI355X_MICROARCH.md, HBM section: gfx950 tallies a 128-B read request as 64 B).  (None, reason) if
    rocprofv3 is missing or a pass fails."""
    kib = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        got, why = rocprof_pass([counter], argv_tail)
        if got is None:
            return None, why
        kib[counter] = got[counter]
    return (2 * kib["FETCH_SIZE"] + kib["WRITE_SIZE"]) * 1024.0, \
        "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over this workload in this run; 2*FETCH+WRITE KiB"


def measure_pipe_counters(argv_tail, n_xcd=8):
    """A third pass: how busy the matrix pipe was and at which clock, from the SQ / GRBM counters of the
    same launches (MI355X_MICROARCH.md: SQ_WAVE_CYCLES counts quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles;
    effective clock = GRBM_GUI_ACTIVE per XCD / kernel time).  None if the pass fails."""
    counters = ["GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA",
                "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAVES"]
    got, why = rocprof_pass(counters, argv_tail)
    if got is None:
        return {"error": why}
    mfma = got["SQ_INSTS_MFMA"]
    return {
        "source": "rocprofv3 --kernel-trace --pmc " + " ".join(counters) + " over this workload in this run (profiled "
                  "launches clock a few % lower than the timed ones)",
        "kernel_us_profiled": got["duration_ns"] * 1e-3,
        "effective_clock_mhz": got["GRBM_GUI_ACTIVE"] / n_xcd / got["duration_ns"] * 1e3,
        "mfma_busy_frac_of_wave_time": got["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * got["SQ_WAVE_CYCLES"]),
        "wave_cycles_per_mfma": 4.0 * got["SQ_WAVE_CYCLES"] / mfma,
        "valu_insts_per_mfma": (got["SQ_INSTS_VALU"] - mfma) / mfma,   # SQ_INSTS_VALU counts the MFMAs too
        "lds_insts_per_mfma": got["SQ_INSTS_LDS"] / mfma,
        "salu_insts_per_mfma": got["SQ_INSTS_SALU"] / mfma,
        "mfma_insts": mfma,
        "waves": got["SQ_WAVES"],
    }


# ---- CPU baseline -----------------------------------------------------------------------------
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


# ---- host side of an N-rank run: where each rank's launcher thread runs ---------------------------
def _parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def _numa_nodes():
    """{node: [cpus]} from /sys/devices/system/node (intersected with what this process may run on); one node 0 with every
    allowed CPU where the box exposes none."""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    nodes = {}
    base = "/sys/devices/system/node"
    try:
        for name in sorted(os.listdir(base)):
            if name.startswith("node") and name[4:].isdigit():
                cpus = [c for c in _parse_cpulist(open(os.path.join(base, name, "cpulist")).read()) if c in set(allowed)]
                if cpus:
                    nodes[int(name[4:])] = cpus
    except OSError:
        pass
    return nodes or {0: allowed}


def _gpu_numa_node(device_index):
    """NUMA node of GPU `device_index` (its PCI function's numa_node), or None."""
    try:
        bdf = torch.cuda.get_device_properties(device_index).pci_bus_id if hasattr(torch.cuda.get_device_properties(device_index), "pci_bus_id") else None
    except Exception:  # noqa: BLE001
        bdf = None
    cands = []
    if bdf:
        cands.append(f"/sys/bus/pci/devices/{str(bdf).lower()}/numa_node")
    hw = hwmon_dir(device_index)
    if hw:
        cands.append(os.path.join(os.path.dirname(os.path.dirname(hw)), "numa_node"))
    for path in cands:
        try:
            node = int(open(path).read().strip())
            if node >= 0:
                return node
        except (OSError, ValueError):
            continue
    return None


def _format_cpulist(cpus):
    """[0, 1, 2, 3, 8, 10, 11] -> '0-3,8,10-11'."""
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def pin_rank(rank, world, local_rank, gpu_nodes=None):
    """Pin this rank's process to its own slice of the CPUs of its GPU's NUMA node (the only resource N ranks
    share is the host -- 13-14 us of launch work per forward and rank, which should neither migrate between sockets nor
    queue behind another rank's thread).  `gpu_nodes`: the NUMA node of every rank's GPU, in rank order (None entries, or
    no list at all -- a dry run --, for "unknown").  The ranks whose GPUs sit on one node split that node's CPUs evenly, in
    rank order; ranks without NUMA information split every allowed CPU the same way.  -> what was done, for the JSON line."""
    nodes = _numa_nodes()
    gpu_nodes = list(gpu_nodes) if gpu_nodes else [None] * world
    mine_node = gpu_nodes[rank] if rank < len(gpu_nodes) else None
    if mine_node in nodes:
        peers = [r for r in range(world) if r < len(gpu_nodes) and gpu_nodes[r] == mine_node]
        cpus, source = nodes[mine_node], "gpu numa_node"
    else:
        peers = [r for r in range(world) if not (r < len(gpu_nodes) and gpu_nodes[r] in nodes)]
        cpus, source = sorted(c for cs in nodes.values() for c in cs), "no NUMA information for the device: CPUs split by rank"
        mine_node = None
    slot, per = peers.index(rank), len(peers)
    n = max(1, len(cpus) // per)
    mine = cpus[slot * n:(slot + 1) * n] or cpus
    info = {"rank": rank, "numa_node": mine_node, "cpus": _format_cpulist(mine), "n_cpus": len(mine),
            "ranks_on_this_node": per, "source": source}
    # every thread the process already has (torch and HIP start workers at import / first use, and sched_setaffinity(0)
    # moves the calling thread only -- ADVICE r05), then the process default for the threads still to come
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = []
    moved, failed = 0, 0
    for tid in tids:
        try:
            os.sched_setaffinity(tid, mine)
            moved += 1
        except (AttributeError, OSError):
            failed += 1
    try:
        os.sched_setaffinity(0, mine)
        info["pinned"] = True
    except (AttributeError, OSError) as exc:
        info["pinned"] = False
        info["error"] = repr(exc)
    info["threads_pinned"], info["threads_not_pinned"] = moved, failed
    try:
        info["affinity_now"] = _format_cpulist(sorted(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    return info


def rocprof_command(rank, out_dir, argv):
    """--rocprof-rank R: the command rank R replaces itself with -- the same bench process under `rocprofv3 --kernel-trace
    --stats` (its own run: no counters, so nothing gpurun refuses), writing <out_dir>/scale_rank<R>_*.csv."""
    return ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out_dir, "-o", f"scale_rank{rank}", "--",
            sys.executable, os.path.abspath(__file__)] + list(argv)


# ---- N > 1 without an external launcher ---------------------------------------------------------
def self_launch(n):
    """`python bench.py --gpus N` with no RANK in the environment: start one rank per GPU (the same
    command, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), wait for all of
    them; rank 0 prints the line.  Returns the worst exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def run_c2_sweep(args, device):
    """--workload c2: one JSON line whose value is the harmonic mean over the sweep."""
    import flash_attention
    from flash_helpers import kernel_configs as kc

    per_s = {}
    hw = hwmon_dir(device.index or 0)
    sweep = [(s_, b_) for s_, b_ in C2_SWEEP if not args.c2_shape or s_ == args.c2_shape]
    for seq, batch in sweep:
        cfg = kc.parse_kernel_name_into_config(args.kernel) if args.kernel else kc.best_config(kc.DType.BF16, seq)
        gen = torch.Generator(device=device).manual_seed(seq)
        q, k, v = (torch.randn((batch, seq, 16, 128), dtype=torch.bfloat16, device=device, generator=gen)
                   for _ in range(3))
        o = torch.empty_like(q)
        if args.traffic_child:  # under rocprofv3: a few launches of this one shape, nothing else
            for _ in range(3):
                flash_attention.forward(cfg, q, k, v, o)
            torch.cuda.synchronize(device)
            return
        # wake the clocks (see --precondition-ms), per shape: allocation and randn idle the chip
        precondition(lambda: flash_attention.forward(cfg, q, k, v, o), device, args.precondition_ms)
        sampler = ClockSampler(hw)
        sampler.start()   # (before the warm-ups: see main())
        for _ in range(args.warmup):
            flash_attention.forward(cfg, q, k, v, o)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            flash_attention.forward(cfg, q, k, v, o)
        torch.cuda.synchronize(device)
        t_end = time.perf_counter()
        sec = (t_end - t0) / args.steps
        sampler.stop()
        clk = sampler.summary(t0, t_end)
        flop = mfma_flop(batch, 16, seq, 128)
        alg_bytes = 4 * batch * seq * 16 * 128 * 2
        # which roof bounds THIS shape: arithmetic intensity (= seq_len / 2 FLOP/B here) against the chip's balance
        # PEAK / HBM = 2500 TFLOP/s / 8 TB/s = 312.5 FLOP/B -- S = 512 sits below it (HBM-bound: 16.8 us at 8 TB/s =
        # 2048 TFLOP/s-equivalent), every longer shape above
        ai = flop / alg_bytes
        hbm_equiv = ai * HBM_PEAK_GBPS * 1e9 / 1e12
        bound = "hbm" if hbm_equiv < PEAK_TFLOPS["bf16"] else "mfma"
        roof = min(hbm_equiv, PEAK_TFLOPS["bf16"])
        tf = flop / sec / 1e12
        per_s[seq] = {"tflops": tf, "ms": sec * 1e3,
                      "batch": batch, "kernel": cfg.short_form(),
                      "sclk_mhz": clk.get("sclk_mhz", {}).get("mean"), "power_w": clk.get("power_w", {}).get("mean"),
                      "algorithmic_bytes": alg_bytes,
                      "roofline": {"bound": bound, "arithmetic_intensity_flop_per_byte": ai,
                                   "achieved": (alg_bytes / sec / 1e9 if bound == "hbm" else tf),
                                   "peak": (HBM_PEAK_GBPS if bound == "hbm" else PEAK_TFLOPS["bf16"]),
                                   "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
                                   "frac": tf / roof, "roof_tflops_equivalent": roof}}
        if not args.no_traffic:  # HBM bytes per launch of this shape, measured (two rocprofv3 PMC passes)
            traffic, _how = measure_traffic(["--workload", "c2", "--c2-shape", str(seq)]
                                            + (["--kernel", args.kernel] if args.kernel else []))
            per_s[seq]["traffic"] = traffic
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
                     "frac": value / PEAK_TFLOPS["bf16"],
                     "bound_per_seq_len": {str(s_): r["roofline"]["bound"] for s_, r in per_s.items()},
                     "note": "the harmonic mean against the MFMA roof; each shape's own roof (S = 512 is HBM-bound: its "
                             "intensity S/2 = 256 FLOP/B is below the chip's 312.5) and fraction are under per_seq_len",
                     "harmonic_mean_of_roofs": statistics.harmonic_mean([r["roofline"]["roof_tflops_equivalent"] for r in per_s.values()]),
                     "traffic": (sum(r["traffic"] for r in per_s.values())
                                 if all(r.get("traffic") for r in per_s.values()) else None),
                     "algorithmic_bytes": sum(r["algorithmic_bytes"] for r in per_s.values()),
                     "traffic_source": "sum over the six shapes of 2*FETCH_SIZE + WRITE_SIZE per launch, rocprofv3 PMC passes "
                                       "in this run (per shape under per_seq_len)"},
    }), flush=True)


def dry_run(args, rank, world):
    """--cpu-dry-run: the control flow of main() around a step that sleeps (no kernel, no oracle)."""
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():
            dist.init_process_group("gloo")
            dist.barrier()
        barrier = dist.barrier
    else:
        def barrier():
            return None
    if args.steps is None:
        args.steps = 5
    affinity = pin_rank(rank, world, int(os.environ.get("LOCAL_RANK", "0"))) if (world > 1 and not args.no_pin) else None
    _, batch, heads, seq, d = WORKLOADS[args.workload if args.workload != "c2" else "c1"]
    lo, hi = shard_for_rank(batch * world, world, rank)
    seconds = timed_steps(lambda: time.sleep(0.002), args.steps, args.warmup, lambda: None, barrier)
    mine = mfma_flop(hi - lo, heads, seq, d) * args.steps / seconds / 1e12
    seconds = max_over_ranks(seconds, world, torch.device("cpu"))
    sustained = sustained_region(lambda: time.sleep(0.002), lambda: None, barrier, seconds, args.steps, world, torch.device("cpu"))
    per_rank = [mine]
    affinities = [affinity]
    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([mine], dtype=torch.float64)
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        per_rank = [float(g.item()) for g in got]
        affinities = [None] * world
        dist.all_gather_object(affinities, affinity)
    if rank == 0:
        value = mfma_flop(hi - lo, heads, seq, d) * world * args.steps / seconds / 1e12
        print(json.dumps({"metric": "cpu-dry-run (a step is a 2 ms sleep: NOT a measurement)", "value": value,
                          "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": seconds / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "none (dry run)",
                          "config": {"workload": f"dry run of {args.workload}", "global_batch": batch * world, "heads": heads,
                                     "seq_len": seq, "flop_per_step_per_gpu": mfma_flop(hi - lo, heads, seq, d),
                                     "shards": [list(shard_for_rank(batch * world, world, r)) for r in range(world)]},
                          "per_gpu_tflops": per_rank,
                          "per_gpu": [{"affinity": a} for a in (affinities if world > 1 else [affinity])],
                          "rocprof_rank": (args.rocprof_rank if args.rocprof_rank >= 0 else None),
                          "rocprof_command": (rocprof_command(args.rocprof_rank, args.rocprof_dir, [a for a in sys.argv[1:] if a != "--cpu-dry-run"])
                                              if args.rocprof_rank >= 0 else None),
                          "sustained": (dict(sustained, tflops=mfma_flop(hi - lo, heads, seq, d) * world * sustained["steps"]
                                             / sustained["seconds"] / 1e12) if sustained else None)}), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


def precondition(step, device, ms, sync_every_batch=False, batch=16):
    """Untimed launches of `step` for `ms` milliseconds: the clock governor needs load to leave its idle state.  The device is
    kept CONTINUOUSLY busy -- a batch is enqueued while the one before still runs (an event behind every batch, the host
    waits for the batch before the last) -- because a synchronize after every few launches (rounds 2-3: every 8) idles the
    chip for ~0.1 ms each time and the governor then never settles where a long run does.  Returns (steps, seconds per step
    of the last completed batch).  FA_BENCH_PRECONDITION_SYNC=1 restores the old loop for an A/B."""
    sync_every_batch = sync_every_batch or os.environ.get("FA_BENCH_PRECONDITION_SYNC") == "1"
    stream = torch.cuda.current_stream(device)
    n, t_start, per_step = 0, time.perf_counter(), 0.0
    if sync_every_batch:
        while (time.perf_counter() - t_start) * 1e3 < ms or n < 8:
            t_a = time.perf_counter()
            for _ in range(8):
                step()
            torch.cuda.synchronize(device)
            per_step = (time.perf_counter() - t_a) / 8
            n += 8
        return n, per_step
    pending = []
    while (time.perf_counter() - t_start) * 1e3 < ms or n < 8:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(batch):
            step()
        e1.record(stream)
        n += batch
        pending.append((e0, e1))
        if len(pending) > 2:
            a0, a1 = pending.pop(0)
            a1.synchronize()                     # (the device still has two batches queued behind this one)
            per_step = a0.elapsed_time(a1) * 1e-3 / batch
    return n, per_step


def _timed_protocol(forward, cfg_, q, k, v, o, steps, warmup, pre_ms, sync):
    """The protocol of `value` in small: `pre_ms` of untimed launches (the chip is warm behind the main region: 100 ms
    re-settles it after a variant switch), `warmup` launches, then `steps` launches between two synchronisations."""
    precondition(lambda: forward(cfg_, q, k, v, o), q.device, pre_ms)
    for _ in range(warmup):
        forward(cfg_, q, k, v, o)
    sync()
    t_a = time.perf_counter()
    for _ in range(steps):
        forward(cfg_, q, k, v, o)
    sync()
    return (time.perf_counter() - t_a) / steps


def _interleaved(named_cfgs, q, k, v, o, args, flop, sync, rounds=5):
    """Every variant once per round, `rounds` rounds, the FIRST ROUND DROPPED (it follows whatever ran before and idled the
    chip); per variant the mean / min / max over the kept rounds, and against the first variant the per-round ratio
    (neighbours in time) as mean and range -- a ratio's range says whether it is a difference or the box."""
    import flash_attention

    per = {name: [] for name, _ in named_cfgs}
    for r in range(rounds):
        for name, c in named_cfgs:
            try:
                sec = _timed_protocol(flash_attention.forward, c, q, k, v, o, args.steps, args.warmup,
                                      min(args.precondition_ms, 100.0), sync)
            except RuntimeError as exc:
                per[name] = {"error": str(exc)[:200]}
                continue
            if r > 0 and isinstance(per[name], list):
                per[name].append(flop / sec / 1e12)
    base = named_cfgs[0][0]
    out = {}
    for name, c in named_cfgs:
        vals = per[name]
        if not isinstance(vals, list) or not vals:
            out[name] = vals if isinstance(vals, dict) else {"error": "no measurement"}
            continue
        rec = {"tflops": statistics.mean(vals), "tflops_min": min(vals), "tflops_max": max(vals), "rounds_kept": len(vals),
               "kernel": c.short_form()}
        if name != base and isinstance(per[base], list) and len(per[base]) == len(vals):
            ratios = [a / b for a, b in zip(vals, per[base])]
            rec["ratio_to_" + base] = {"mean": statistics.mean(ratios), "min": min(ratios), "max": max(ratios)}
        out[name] = rec
    out["protocol"] = (f"{rounds} rounds of every variant in turn (each: <= 100 ms of untimed launches, {args.warmup} warm-ups, "
                       f"{args.steps} timed launches between two synchronisations), first round dropped; ratios per round")
    return out


def side_by_side(cfg, q, k, v, o, args, flop, sync):
    """`variants` of the driver line: the default kernel (the stateless speculative softmax since round 6) beside (a) the
    running-max (lazy rescale) kernel -- the reference's arithmetic family, north_star's "fp32 running max/sum" literally;
    its number is also the line's top-level `value_reference_arithmetic` --, (b) the opt-in adaptive mode of rounds 4-5 (on
    benign data the same kernel as the default) and (c) the opt-in pre-scaled Q (NOT the reference's arithmetic: DESIGN.md
    3.7; never `value`)."""
    from dataclasses import replace as _replace

    named = [("default", cfg)]
    if getattr(cfg, "speculative_softmax", False):
        named.append(("lazy", _replace(cfg, speculative_softmax=False, adaptive_softmax=False)))
        if not getattr(cfg, "adaptive_softmax", False):
            named.append(("adaptive_opt_in", _replace(cfg, adaptive_softmax=True)))
        else:
            named.append(("speculative_always", _replace(cfg, adaptive_softmax=False)))
    named.append(("prescaled_q", _replace(cfg, prescaled_q=True, adaptive_softmax=False)))
    out = _interleaved(named, q, k, v, o, args, flop, sync)
    if isinstance(out.get("prescaled_q"), dict) and "tflops" in out["prescaled_q"]:
        out["prescaled_q"]["note"] = ("opt-in (fa_fwd_opts.prescaled_q): Q * log2(e)/sqrt(d) rounded to 16 bit once instead of an "
                                      "fp32 multiply per logit; inside the reference's tolerance rule on benchmark-like data "
                                      "(profiles/r03/prescaled_q_error.txt), not its arithmetic: NOT `value`")
    return out


def robustness(cfg, shape, dtype, device, args, flop, sync):
    """`robustness` of the driver line: the same workload shape on data that is not N(0, 1) (make_inputs: `heavy` = Student-t
    K, `sink` = +12 nats on the first four keys -- which the speculative first pass visits FIRST since round 6), the
    always-speculative kernel (`speculative_always`: what best_config() is since round 6, so `default` is the same kernel)
    beside the running-max one (`lazy`) and the opt-in adaptive mode, interleaved; `items_redone` of one speculative launch;
    what the adaptive mode did."""
    from dataclasses import replace as _replace

    import flash_attention
    import flash_attention_kernels
    from flash_attention_from_scratch_amd import _capi

    if not getattr(cfg, "speculative_softmax", False):
        return None
    spec = _replace(cfg, adaptive_softmax=False)
    lazy = _replace(cfg, speculative_softmax=False, adaptive_softmax=False)
    adaptive = _replace(cfg, adaptive_softmax=True)
    out = {}
    cases = [("heavy", dtype), ("sink", dtype)]
    if dtype == torch.bfloat16:
        cases.append(("sink", torch.float16))   # fp16's 16-bit P leaves ~10 nats: the case that kept round 3's fp16 default lazy
        cases.append(("heavy", torch.float16))  # ... and the one the stateless default does NOT cover: most items run twice
    for data, dt in cases:
        gen = torch.Generator(device=device).manual_seed(4242)
        q, k, v = make_inputs(data, shape, dt, device, gen)
        o = torch.empty_like(q)
        from flash_helpers import kernel_configs as kc

        name = kc.DType.BF16 if dt == torch.bfloat16 else kc.DType.FP16
        c_def, c_spec, c_lazy, c_ada = (_replace(c, dtype=name) for c in (cfg, spec, lazy, adaptive))
        _capi.adaptive_reset(device.index or 0)
        before = _capi.adaptive_state(device.index or 0)
        stats = torch.zeros(2, dtype=torch.int32, device=device)
        flash_attention_kernels.forward(c_spec, q, k, v, o, stats=stats)
        sync()
        items, redone = (int(x) for x in stats.tolist())
        rec = _interleaved([("lazy", c_lazy), ("default", c_def), ("speculative_always", c_spec), ("adaptive_opt_in", c_ada)],
                           q, k, v, o, args, flop, sync, rounds=4)
        after = _capi.adaptive_state(device.index or 0)
        rec["items"] = items
        rec["items_redone"] = redone
        rec["items_redone_by_an_always_speculative_launch"] = redone
        if isinstance(rec.get("speculative_always"), dict):
            rec["speculative_always"]["items_redone"] = redone
        rec["adaptive"] = {"launches": after["launches"] - before["launches"], "demoted": after["demoted"],
                           "reports": after["reports"], "hold": after["hold"]}
        rec["default_over_lazy"] = (rec["default"].get("ratio_to_lazy", {}).get("mean")
                                   if isinstance(rec.get("default"), dict) else None)
        out[data + ("_fp16" if dt != dtype else "")] = rec
        del q, k, v, o
    _capi.adaptive_reset(device.index or 0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (exactly this many).  Default: 50 at N = 1; at N > 1 as many as make the timed region "
                         ">= 200 ms per rank (a 20-step region is 8 ms, and the max over ranks then mostly measures host jitter)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c1", choices=sorted(WORKLOADS) + ["c2"])
    ap.add_argument("--kernel", default="", help="short-form config; default = best_config(dtype)")
    ap.add_argument("--dtype", default="", choices=["", "bf16", "fp16"], help="override the workload's dtype (same shape)")
    ap.add_argument("--data", default="randn", choices=["randn", "sink", "heavy"],
                    help="synthetic inputs: N(0,1) (the reference's; default) | an attention sink (+12 nats at the first 4 keys) "
                         "| heavy-tailed K (Student-t, 3 dof): what the speculative softmax's second pass costs (line: speculative)")
    ap.add_argument("--hermetic-reps", type=int, default=50,
                    help="launches of the side measurement under the reference's flush + idle-spin protocol (protocols.hermetic); 0 = skip")
    ap.add_argument("--no-mfma-roof", action="store_true", help="skip the register-only MFMA loop (roofline.mfma_only_*)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the side-by-side variants and the robustness block")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the three rocprofv3 PMC passes (HBM bytes per launch; matrix-pipe occupancy, "
                         "instruction mix and effective clock) (~1 min)")
    ap.add_argument("--precondition-ms", type=float, default=400.0,
                    help="untimed launches of the same step for this long BEFORE the W warm-up steps: the chip's "
                         "clock governor needs a few hundred ms of load to leave its idle state (a cold 20-step "
                         "run reads ~12 %% low, DESIGN.md 5); 0 disables; stated in the line")
    ap.add_argument("--hermetic", action="store_true",
                    help="every timed launch on its own behind a cache flush + idle spin (pt_bench protocol)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--c2-shape", type=int, default=0, help=argparse.SUPPRESS)  # traffic child of the c2 sweep: one seq_len
    ap.add_argument("--cpu-dry-run", action="store_true",
                    help="no GPU, no kernel: a step is a 2 ms sleep.  Exercises the launcher, the shard arithmetic, "
                         "the barrier-bracketed timing and the rank-0 line on a CPU-only box (tests/); the line says so")
    ap.add_argument("--no-pin", action="store_true", help="N > 1: do not pin the ranks to the NUMA nodes of their GPUs")
    ap.add_argument("--rocprof-rank", type=int, default=-1,
                    help="N > 1: rank R runs under `rocprofv3 --kernel-trace --stats` and leaves --rocprof-dir/scale_rank<R>*.csv "
                         "(one rank only: the profiled rank clocks a few % lower, see MI355X_MICROARCH.md DVFS)")
    ap.add_argument("--rocprof-dir", default=os.path.join("profiles", "r06"))
    ap.add_argument("--batch-per-rank", type=int, default=0,
                    help="override the workload's per-GPU batch (weak scaling keeps it fixed per rank).  `--gpus 8 --workload c4 "
                         "--batch-per-rank 1` self-launched on a ONE-GPU box is the eight-launchers-one-host rehearsal "
                         "(profiles/r06/bench_n8_selflaunch.json): what it measures is the host -- per-rank enqueue time, "
                         "affinity, barrier latency --, not the device")
    ap.add_argument("--dist-backend", default="gloo", choices=["gloo", "nccl"],
                    help="barrier + max-over-ranks only (the data path has no collective): gloo (default; on "
                         "a box with fewer GPUs than ranks the ranks share devices and the timings mean "
                         "nothing) or nccl (= RCCL)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")

    if args.rocprof_rank == rank and world > 1 and not args.cpu_dry_run and not os.environ.get("FA_UNDER_ROCPROF"):
        os.environ["FA_UNDER_ROCPROF"] = "1"   # (the re-executed process keeps RANK / WORLD_SIZE / MASTER_*: it joins the same rendezvous)
        os.makedirs(args.rocprof_dir, exist_ok=True)
        os.execvp("rocprofv3", rocprof_command(rank, args.rocprof_dir, sys.argv[1:]))

    if args.cpu_dry_run:
        return dry_run(args, rank, world)

    import flash_attention
    from flash_helpers import kernel_configs as kc

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev and args.dist_backend != "gloo":
        raise SystemExit(f"rank {rank}: no device {local_rank} (nccl needs one GPU per rank)")
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    affinity = None
    reduce_device = device if args.dist_backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():  # (the transport's banner goes to stderr: stdout is the JSON line's)
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", device_id=device)  # RCCL; barrier + max only
            else:
                dist.init_process_group("gloo")
            dist.barrier()
        if not args.no_pin:
            # each rank's ACTUAL device, gathered (not assumed from the rank number: ADVICE r05), then its NUMA node
            devs = [None] * world
            dist.all_gather_object(devs, local_rank)
            affinity = pin_rank(rank, world, local_rank, [_gpu_numa_node(d_) for d_ in devs])
            affinity["device"] = local_rank

    if args.workload == "c2":
        if world != 1:
            raise SystemExit("the c2 sweep is a single-GPU workload")
        if args.steps is None:
            args.steps = 50
        return run_c2_sweep(args, device)
    dtype_name, batch, heads, seq, d = WORKLOADS[args.workload]
    if args.batch_per_rank > 0:
        batch = args.batch_per_rank
    if args.dtype:
        dtype_name = args.dtype
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[dtype_name]
    cfg = (kc.parse_kernel_name_into_config(args.kernel) if args.kernel
           else kc.best_config(kc.DType.BF16 if dtype_name == "bf16" else kc.DType.FP16, seq))

    # this rank's shard of the global batch (weak scaling: `batch` per GPU)
    lo, hi = shard_for_rank(batch * world, world, rank)
    gen = torch.Generator(device=device).manual_seed(1000 + rank)
    slab = torch.empty((4, hi - lo, seq, heads, d), dtype=dtype, device=device)
    q, o, k, v = slab[0], slab[1], slab[2], slab[3]   # generate_qkvo layout (utils.py:124-134)
    for dst, src in zip((q, k, v), make_inputs(args.data, (hi - lo, seq, heads, d), dtype, device, gen)):
        dst.copy_(src)
        del src

    stream = torch.cuda.current_stream(device)

    def step():
        flash_attention.forward(cfg, q, k, v, o)

    def sync():
        torch.cuda.synchronize(device)

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()

    if args.traffic_child:  # under rocprofv3 (measure_traffic): a few launches of the workload, nothing else
        for _ in range(3):
            step()
        sync()
        return

    flush_buf = torch.empty(512 * 1024 * 1024, dtype=torch.int8, device=device) if (args.hermetic or args.hermetic_reps) else None

    def hermetic_launch():
        """pt_bench.py:145-174: flush (> L2 + Infinity Cache), idle spin, events around ONE launch -> ms"""
        flush_buf.zero_()
        torch.cuda._sleep(1_000_000)
        sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        step()
        e1.record(stream)
        sync()
        return e0.elapsed_time(e1)

    sampler = ClockSampler(hwmon_dir(local_rank))
    # wake the clocks: the same launches, untimed, until --precondition-ms have passed
    pre_steps, pre_step_s = precondition(step, device, args.precondition_ms)
    if args.steps is None:
        # N = 1: 50.  N > 1: a timed region of >= 200 ms per rank (from the last preconditioning batch's step time)
        if pre_step_s <= 0.0:   # (a preconditioning too short to complete a batch: one timed batch)
            sync()
            t_a = time.perf_counter()
            for _ in range(8):
                step()
            sync()
            pre_step_s = (time.perf_counter() - t_a) / 8
        est = max(pre_step_s, 1e-6)
        args.steps = 50 if world == 1 else max(50, int(0.2 / est) + 1)
        if world > 1:  # every rank times the same number of steps
            import torch.distributed as dist

            t = torch.tensor([args.steps], dtype=torch.int64, device=reduce_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            args.steps = int(t.item())

    # events on the launch stream (flash_attention launches on torch's current stream): one in front
    # of the first timed launch and one behind every launch
    events = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    herm_ms = []

    def timed_step(i):
        if args.hermetic:
            herm_ms.append(hermetic_launch())
            return
        if i == 0:
            events[0].record(stream)
        step()
        events[i + 1].record(stream)

    late = os.environ.get("FA_BENCH_SAMPLER_LATE") == "1"   # (A/B of this ordering: profiles/r04/sampler_start.txt)
    if not late:
        sampler.start()   # (before the warm-ups: nothing but the bracket itself sits between them and the first timed launch)
    for _ in range(args.warmup):
        step()
    sync()
    barrier()
    if late:
        sampler.start()   # round 3's order: the thread's start-up (~0.1-1 ms of an idle GPU) in front of the first launch
    t0 = time.perf_counter()
    for i in range(args.steps):
        timed_step(i)
    t_enqueued = time.perf_counter()   # (the host's share of the region: K launches enqueued -- SURVEY 8e's shared resource)
    barrier()   # (host side, while the devices still work: see timed_steps)
    t_barrier = time.perf_counter()
    sync()
    t_end = time.perf_counter()
    seconds = t_end - t0
    sampler.stop()
    if args.hermetic:
        per_launch = herm_ms
        seconds = sum(herm_ms) * 1e-3  # the flushes and spins are not steps
        kernel_ms = statistics.mean(herm_ms)
    else:
        per_launch = [events[i].elapsed_time(events[i + 1]) for i in range(args.steps)]
        kernel_ms = events[0].elapsed_time(events[-1]) / args.steps  # avg launch duration, this rank
    seconds = max_over_ranks(seconds, world, reduce_device)
    sustained = None if args.hermetic else sustained_region(step, sync, barrier, seconds, args.steps, world, reduce_device)

    flop_per_step_rank = mfma_flop(hi - lo, heads, seq, d)
    total_flop = flop_per_step_rank * world * args.steps  # equal shards
    value = total_flop / seconds / 1e12
    achieved = flop_per_step_rank / (kernel_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[dtype_name]

    # one more launch, untimed, with the device-side statistics on: how many items the speculative softmax ran twice
    stats = torch.zeros(2, dtype=torch.int32, device=device)
    import flash_attention_kernels

    flash_attention_kernels.forward(cfg, q, k, v, o, stats=stats)
    sync()
    items, redone = (int(x) for x in stats.tolist())
    adaptive_record = None
    if getattr(cfg, "adaptive_softmax", False):  # what the adaptive mode did over this process's launches on the device
        from flash_attention_from_scratch_amd import _capi

        st = _capi.adaptive_state(local_rank)
        adaptive_record = {k_: st[k_] for k_ in ("launches", "demoted", "reports", "hold", "mode")}

    per_rank = per_rank_clocks = None
    if world > 1:  # per-GPU rates, clocks and power next to the aggregate (rank order)
        import torch.distributed as dist

        clk = sampler.summary(t0, t_end)
        import hashlib

        # barrier latency with nothing else going on: five back-to-back gloo / RCCL barriers, the median
        lat = []
        for _ in range(5):
            tb = time.perf_counter()
            dist.barrier()
            lat.append((time.perf_counter() - tb) * 1e6)
        mine = {"tflops": achieved, "sclk_mhz": clk.get("sclk_mhz", {}).get("mean"), "power_w": clk.get("power_w", {}).get("mean"),
                "device": local_rank, "kernel_ms": kernel_ms, "affinity": affinity,
                "under_rocprof": bool(os.environ.get("FA_UNDER_ROCPROF")),
                # what N launchers share is the HOST: this rank's enqueue time per launch inside the timed region, how long
                # its closing barrier took (host side, devices still working), an idle barrier's latency
                "host_us_per_launch": (t_enqueued - t0) / args.steps * 1e6,
                "closing_barrier_us": (t_barrier - t_enqueued) * 1e6,
                "idle_barrier_us_median": statistics.median(lat),
                "batch_rows": [lo, hi],
                # this rank's output of the LAST step (inputs seeded 1000 + rank: a single process can reproduce every shard)
                "output_sha256": hashlib.sha256(o.cpu().view(torch.int16).numpy().tobytes()).hexdigest()}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = [g["tflops"] for g in gathered]
        per_rank_clocks = gathered

    if rank == 0:
        props = torch.cuda.get_device_properties(device)
        clocks = sampler.summary(t0, t_end)
        sclk = clocks.get("sclk_mhz", {}).get("mean")
        line = {
            "metric": "achieved bf16 TFLOPs and % of MFMA peak at seq_len=4096 d_head=128"
                      if (args.workload == "c1" and dtype_name == "bf16") else f"achieved {dtype_name} TFLOPs ({args.workload})",
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
            "data": "synthetic" if args.data == "randn" else f"synthetic ({args.data}: see bench.py make_inputs)",
            "pct_of_mfma_peak": 100.0 * value / (peak * world),
            "ref_convention_tflops": value * (4 * d + 6) / (4 * d),  # B*H*(4S^2d+6S^2), kernel_configs.py:102
            "protocol": "hermetic: flush + idle spin before every launch, value from the per-launch events"
                        if args.hermetic else "back-to-back launches",
            "precondition": {"ms": args.precondition_ms, "untimed_steps": pre_steps,
                             "why": "clock governor ramp from idle; before the W warm-up steps, outside the timed region; the "
                                    "device is kept continuously busy (no synchronize between the batches)"},
            "config": {
                "workload": f"{args.workload}: FA2 forward {dtype_name} batch={batch}/GPU heads={heads} "
                            f"seq_len={seq} d_head={d} non-causal",
                "global_batch": batch * world,
                "kernel": cfg.short_form(),
                "softmax_mode": kc.softmax_mode(cfg),
                # (round 6: > 0 = the launcher takes the form whose every second round of a long head walks K / V
                # [tile 0, then last-to-second]; the value is the workgroups per XCD -- Q block qb walks that way when (qb // G) is odd)
                "kv_walk_alternates": kc.kv_walk_alternates(cfg, (hi - lo) * heads, seq, num_cus=props.multi_processor_count),
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
                "traffic": None,
                "algorithmic_bytes": 4 * (hi - lo) * seq * heads * d * 2,
                "kernel_ms": kernel_ms,
                "kernel_ms_per_launch": dict(distribution(per_launch), first=per_launch[0],
                                             argmax=max(range(len(per_launch)), key=per_launch.__getitem__)),
                "flop_per_launch": flop_per_step_rank,
                # the matrix pipe's rate at the clock the chip actually held (it clocks to its power budget)
                "peak_at_measured_clock": peak * sclk / MAX_SCLK_MHZ if sclk else None,
                "frac_of_peak_at_measured_clock": achieved / (peak * sclk / MAX_SCLK_MHZ) if sclk else None,
            },
            "clocks": clocks,
            "speculative": {"items": items, "items_redone": redone, "second_pass_fraction": redone / items if items else None,
                            "source": "fa_fwd_stats of one more (untimed) launch of the same step",
                            "adaptive": adaptive_record},
        }
        if sustained:
            sustained["tflops"] = flop_per_step_rank * world * sustained["steps"] / sustained["seconds"] / 1e12
            line["sustained"] = sustained
        # the box-independent trackers, at the top level: how close to the matrix pipe's rate AT THE CLOCK THE
        # CHIP HELD, and (below, once the counter pass has run) the wave cycles one MFMA costs over the whole launch
        line["frac_of_peak_at_measured_clock"] = line["roofline"]["frac_of_peak_at_measured_clock"]
        if per_rank:
            line["per_gpu_tflops"] = per_rank
            line["per_gpu"] = per_rank_clocks
            line["aggregate_over_sum_of_gpus"] = value / sum(per_rank)
        if world == 1 and args.hermetic_reps and not args.hermetic:
            # the same kernel under the reference's protocol, in the same run (pt_bench.py:145-174)
            for _ in range(3):
                hermetic_launch()
            side = [hermetic_launch() for _ in range(args.hermetic_reps)]
            line["protocols"] = {
                "back_to_back": {"tflops": value, "ms": seconds / args.steps * 1e3, "n": args.steps},
                "hermetic": {"tflops": flop_per_step_rank / (statistics.mean(side) * 1e-3) / 1e12,
                             "tflops_median": flop_per_step_rank / (statistics.median(side) * 1e-3) / 1e12,
                             "ms": distribution(side),
                             "what": "512 MiB flush + idle spin + sync before every launch, events around the launch "
                                     "(tools/benchmark/pt_bench.py:145-174), 3 warm-ups"},
            }
        if world == 1 and not args.kernel and not args.no_variants and hasattr(cfg, "prescaled_q") and not cfg.prescaled_q:
            line["variants"] = side_by_side(cfg, q, k, v, o, args, flop_per_step_rank, sync)
            lz = line["variants"].get("lazy")
            if isinstance(lz, dict) and "tflops" in lz:
                # the reference-arithmetic family (running max, lazy rescale) beside `value`, which is the speculative form
                # (say both numbers whenever the headline is quoted).  Same run, interleaved rounds.
                line["value_reference_arithmetic"] = lz["tflops"]
                line["value_reference_arithmetic_what"] = ("TFLOP/s of the running-max (lazy rescale) kernel " + lz["kernel"]
                                                           + ": variants.lazy of this run; `value` is the speculative softmax")
            line["robustness"] = robustness(cfg, (hi - lo, seq, heads, d), dtype, device, args, flop_per_step_rank, sync)
        if world == 1 and not args.no_mfma_roof:
            del flush_buf
            roof = mfma_only_roof(local_rank)
            line["roofline"]["mfma_only"] = roof
            rand = roof.get(f"{dtype_name}_normal")
            if rand:
                line["roofline"]["mfma_only_random_tflops"] = rand
                line["roofline"]["mfma_only_zero_tflops"] = roof.get(f"{dtype_name}_zeros")
                line["roofline"]["frac_of_mfma_only_random"] = achieved / rand
                line["roofline"]["mfma_only_source"] = ("lib/mfma_energy quick (tools/mfma_energy.hip): register-only "
                                                        "v_mfma_f32_32x32x16 loop in the kernel's issue order, one wave per "
                                                        "SIMD, 256 workgroups, ~20 ms launches, this device, after the timed region")
        if world == 1 and not args.no_traffic:
            tail = ["--workload", args.workload, "--data", args.data] + (["--kernel", args.kernel] if args.kernel else []) \
                + (["--dtype", args.dtype] if args.dtype else [])
            traffic, how = measure_traffic(tail)
            if traffic is None:
                line["roofline"]["traffic"] = committed_traffic(cfg.short_form(), args.workload)
                line["roofline"]["traffic_source"] = f"profiles/traffic_{args.workload}.json ({how})"
            else:
                line["roofline"]["traffic"] = traffic
                line["roofline"]["traffic_source"] = how
                # achieved HBM rate against the chip's peak (north star: "achieved HBM GB/s against the chip's peaks"): the
                # measured bytes of a launch over the kernel's average duration in the timed region
                line["roofline"]["hbm_gb_per_s"] = traffic / (kernel_ms * 1e-3) / 1e9
                line["roofline"]["hbm_peak_gb_per_s"] = HBM_PEAK_GBPS
                line["roofline"]["hbm_frac_of_peak"] = traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
                pc = measure_pipe_counters(tail)
                line["roofline"]["pipe_counters"] = pc
                if isinstance(pc, dict) and "wave_cycles_per_mfma" in pc:
                    line["wave_cycles_per_mfma"] = pc["wave_cycles_per_mfma"]
                    line["mfma_busy_frac_of_wave_time"] = pc["mfma_busy_frac_of_wave_time"]
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(dtype, batch, heads, seq, d)
        print(json.dumps(line), flush=True)

    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
