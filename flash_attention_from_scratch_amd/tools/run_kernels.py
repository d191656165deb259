#!/usr/bin/env python3
"""Run selected kernels N times -- the profiling target (rocprofv3), counterpart of
the reference's tools/benchmark/run_kernels.py (there wrapped by Nsight Compute).

    python run_kernels.py 4096 128 --n_runs 5 --kernels "(BF16, 128, 256, 64, 8): async+eager+swizzled+load_0_0_0_tiles"
    KERNELS=native python run_kernels.py 4096 --n_runs 3          # env selector, as the reference
"""
import argparse
import sys

import torch

import flash_attention
from flash_helpers.kernel_configs import DType, get_kernel_configs, parse_kernel_name_into_config
from flash_helpers.test.utils import (
    BATCH_SIZE_FOR_SEQ_LEN,
    BENCHMARK_N_HEADS,
    QKVConfig,
    generate_qkv,
    reference_forward_kernel_v2,
)


def main():
    ap = argparse.ArgumentParser(description="Run flash attention kernels N times")
    ap.add_argument("seq_len", type=int, nargs="?", default=4096)
    ap.add_argument("d_head", type=int, nargs="?", default=128)
    ap.add_argument("--ref", action="store_true", help="also run the torch SDPA comparator")
    ap.add_argument("--n_runs", type=int, required=True)
    ap.add_argument("--kernels", nargs="+", help="kernel config strings (short form or demangled)")
    ap.add_argument("--dtype", type=str, default="BF16", help="dtype of the comparator run")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--heads", type=int, default=BENCHMARK_N_HEADS)
    args = ap.parse_args()
    if args.seq_len <= 0 or args.d_head <= 0 or args.n_runs <= 0:
        print("Error: seq_len, d_head and n_runs must be positive integers")
        sys.exit(1)
    try:
        ref_dtype = DType.from_string(args.dtype)
        selected = ([parse_kernel_name_into_config(s) for s in args.kernels]
                    if args.kernels else get_kernel_configs())
    except ValueError as e:
        print(f"Error: {e}")
        sys.exit(1)

    batch = args.batch or BATCH_SIZE_FOR_SEQ_LEN[args.seq_len]
    tensors = {}
    for dt in (DType.FP16, DType.BF16):
        q, k, v = generate_qkv(QKVConfig(n_heads=args.heads, d_head=args.d_head, batch_size=batch,
                                         seq_len=args.seq_len, dtype=dt.to_torch_dtype(),
                                         device=torch.device("cuda:0")), seed=0)
        tensors[dt] = (q, k, v, torch.empty_like(q))
    torch.cuda.synchronize()

    if args.ref:
        q, k, v, o = tensors[ref_dtype]
        print(f"Running torch SDPA comparator with dtype: {ref_dtype.name}")
        for _ in range(args.n_runs):
            reference_forward_kernel_v2(q, k, v, o)
            torch.cuda.synchronize()
    for cfg in selected:
        if args.seq_len % cfg.B_r or args.seq_len % cfg.B_c:
            continue
        q, k, v, o = tensors[cfg.dtype]
        print(f"Running kernel {cfg.short_form()} with dtype: {cfg.dtype.name}")
        for _ in range(args.n_runs):
            flash_attention.forward(cfg, q, k, v, o)
            torch.cuda.synchronize()


if __name__ == "__main__":
    main()
