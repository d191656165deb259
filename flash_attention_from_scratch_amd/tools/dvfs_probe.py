#!/usr/bin/env python3
"""Same kernel, same instruction stream, different operand data: shows how much of the
headline number is set by the chip's power management rather than by the kernel
(cdna_hip_programming.md 5.4 rule 25).  Round-1 result on MI355X, best config at C1:
random N(0,1) 1074-1105 TFLOP/s, all-zero inputs 1531 TFLOP/s (+40 %)."""
import torch

import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.abspath(_os.path.join(_os.path.dirname(__file__), _os.pardir, _os.pardir)))  # repo root: runs without PYTHONPATH

import flash_attention  # noqa: E402
from flash_helpers import kernel_configs as kc


def main():
    cfg = kc.best_config(kc.DType.BF16, 4096)
    B, S, H, D = 4, 4096, 16, 128

    def run(q, k, v, name):
        o = torch.empty_like(q)
        for _ in range(10):
            flash_attention.forward(cfg, q, k, v, o)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            flash_attention.forward(cfg, q, k, v, o)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        print(f"{name:32s} {ms:.4f} ms  {4 * B * H * S * S * D / ms / 1e9:8.1f} TFLOP/s")

    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = (torch.randn((B, S, H, D), dtype=torch.bfloat16, device="cuda", generator=g) for _ in range(3))
    z = torch.zeros_like(q)
    print(cfg)
    run(q, k, v, "random N(0,1)")
    run(z, z, z, "all zeros")
    run(q, k, z, "random q,k ; zero v")
    run(q * 0.1, k * 0.1, v, "q,k x0.1 (flat softmax)")
    run(q.abs(), k.abs(), v.abs(), "abs (sign bit constant)")
    run(q, k, v, "random N(0,1) again")


if __name__ == "__main__":
    main()
