#!/usr/bin/env python3
"""Profiling target: launch the selected kernels N times each, nothing else -- what rocprofv3
wraps in tools/rocprof_bench.py (the reference wraps its tools/benchmark/run_kernels.py in
Nsight Compute).  Kernels come from --kernels (short-form or demangled names) or, without it,
from the KERNELS environment selector of flash_helpers.kernel_configs.

    python run_kernels.py 4096 128 --n_runs 5 --kernels "(BF16, 128, 256, 64, 8): async+eager+swizzled+load_0_0_0_tiles"
    KERNELS=native python run_kernels.py 4096 --n_runs 3
"""
import argparse

import torch

import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.abspath(_os.path.join(_os.path.dirname(__file__), _os.pardir, _os.pardir)))  # repo root: runs without PYTHONPATH

import flash_attention  # noqa: E402
from flash_helpers import kernel_configs as kc
from flash_helpers.test import utils as ut


def build_inputs(seq_len, d_head, batch, heads):
    """One (q, k, v, o) set per dtype, seeded, on cuda:0."""
    data = {}
    for dt in kc.DType:
        shape = ut.QKVConfig(n_heads=heads, d_head=d_head, batch_size=batch, seq_len=seq_len,
                             dtype=dt.to_torch_dtype(), device=torch.device("cuda:0"))
        q, k, v = ut.generate_qkv(shape, seed=0)
        data[dt] = (q, k, v, torch.empty_like(q))
    torch.cuda.synchronize()
    return data


def main(argv=None):
    ap = argparse.ArgumentParser(description="Launch flash attention kernels N times (profiler target)")
    ap.add_argument("seq_len", type=int, nargs="?", default=4096)
    ap.add_argument("d_head", type=int, nargs="?", default=128)
    ap.add_argument("--n_runs", type=int, required=True)
    ap.add_argument("--kernels", nargs="+", help="kernel config strings; default: KERNELS env selector")
    ap.add_argument("--ref", action="store_true", help="also run the torch SDPA comparator")
    ap.add_argument("--dtype", default="BF16", help="dtype of the comparator run")
    ap.add_argument("--batch", type=int, default=0, help="default: BATCH_SIZE_FOR_SEQ_LEN[seq_len]")
    ap.add_argument("--heads", type=int, default=ut.BENCHMARK_N_HEADS)
    args = ap.parse_args(argv)
    if min(args.seq_len, args.d_head, args.n_runs) <= 0:
        ap.error("seq_len, d_head and n_runs must be positive")
    try:
        configs = [kc.parse_kernel_name_into_config(s) for s in args.kernels] if args.kernels else kc.get_kernel_configs()
        ref_dtype = kc.DType.from_string(args.dtype)
    except ValueError as err:
        ap.error(str(err))

    data = build_inputs(args.seq_len, args.d_head, args.batch or ut.BATCH_SIZE_FOR_SEQ_LEN[args.seq_len], args.heads)
    if args.ref:
        q, k, v, o = data[ref_dtype]
        print(f"torch SDPA comparator, {ref_dtype.name}")
        for _ in range(args.n_runs):
            ut.reference_forward_kernel_v2(q, k, v, o)
            torch.cuda.synchronize()
    for cfg in configs:
        if args.seq_len % cfg.B_r or args.seq_len % cfg.B_c:
            continue  # the reference scope needs tile multiples
        q, k, v, o = data[cfg.dtype]
        print(f"{cfg.short_form()}")
        for _ in range(args.n_runs):
            flash_attention.forward(cfg, q, k, v, o)
            torch.cuda.synchronize()


if __name__ == "__main__":
    main()
