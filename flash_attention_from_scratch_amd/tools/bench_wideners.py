#!/usr/bin/env python3
"""Timing of the scope wideners (causal mask, ragged seq_len) next to the reference-scope
kernel and torch SDPA on the same device.  FLOPs: full = 4BHS^2d; causal = half of that
(the masked upper triangle is not computed: KV tiles above the diagonal are skipped)."""
import torch

import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.abspath(_os.path.join(_os.path.dirname(__file__), _os.pardir, _os.pardir)))  # repo root: runs without PYTHONPATH

import flash_attention  # noqa: E402
from flash_helpers import kernel_configs as kc
from flash_helpers.test import utils as ut


def timed(fn, reps=20, warm_s=0.3, min_s=0.15):
    """Mean ms per call: `warm_s` of untimed calls first (the clock governor needs load to leave idle -- bench.py's
    precondition; with five warm-ups the 0.25-ms causal C1 launches read 5 % low: profiles/r05/wideners.txt vs
    masked_probe.txt before this), then at least `reps` calls and `min_s` seconds between two events."""
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    t0 = time.perf_counter()
    e0.record()
    while n < reps or time.perf_counter() - t0 < min_s:
        for _ in range(reps):
            fn()
        n += reps
        if n >= 16 * reps:
            break
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    print("shape (B,S,H) | mode | kernel ms | TFLOP/s (useful) | torch SDPA ms | TFLOP/s")
    for dtype, name in ((torch.bfloat16, kc.DType.BF16),):
        for B, S, H in ((4, 4096, 16), (4, 4000, 16), (2, 16384, 16), (16, 1000, 16)):
            cfg = kc.best_config(name, S, masked=True)
            qc = ut.QKVConfig(n_heads=H, d_head=128, batch_size=B, seq_len=S, dtype=dtype, device=torch.device("cuda:0"))
            q, k, v = ut.generate_qkv(qc, seed=0)
            o = torch.empty_like(q)
            qt, kt, vt = (t.transpose(1, 2) for t in (q, k, v))
            for causal in (False, True):
                flop = 4 * B * H * S * S * 128 * (0.5 if causal else 1.0)
                ms = timed(lambda: flash_attention.forward_ex(cfg, q, k, v, o, causal=causal))
                ms_ref = timed(lambda: torch.nn.functional.scaled_dot_product_attention(qt, kt, vt, is_causal=causal), reps=8)
                print(f"({B},{S},{H}) | {'causal' if causal else 'full  '} | {ms:8.4f} | {flop / ms / 1e9:8.1f} | "
                      f"{ms_ref:8.4f} | {flop / ms_ref / 1e9:8.1f}   [{cfg.short_form()}]")


if __name__ == "__main__":
    main()
