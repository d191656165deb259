#!/usr/bin/env python3
"""Soak of the persistent kernel's walk with MANY items per workgroup (ordinals beyond 63 share one bit of the failed
mask): short sequences, tens of thousands of (batch, head) pairs, spikes in random items, plain / causal / ragged.
Checked against fp32 attention in chunks of the batch.  Usage: python tools/soak_many_items.py [seconds] [seed]"""
import random
import sys
import time

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import flash_attention  # noqa: E402
from flash_helpers import kernel_configs as kc  # noqa: E402
from soak import eager  # noqa: E402

DEV = "cuda:0"


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        dtype, name = rng.choice(((torch.bfloat16, kc.DType.BF16), (torch.float16, kc.DType.FP16)))
        cfg = kc.NativeKernelConfig(name, 128, 256, 64, 4, True, True, True, 0, 0, 0, True, False, speculative_softmax=rng.random() < 0.8)
        mode = rng.choice(["plain", "causal", "ragged", "ragged-causal"])
        S = rng.choice([256, 512, 768]) if mode in ("plain", "causal") else rng.choice([300, 500, 700])
        H = rng.choice([8, 32, 64, 128, 100])
        B = rng.choice([32, 64, 128, 200])
        gen = torch.Generator(device=DEV).manual_seed(rng.randrange(1 << 30))
        q, k, v = (torch.randn((B, S, H, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        for _ in range(rng.choice([0, 1, 3, 8])):
            b_, h_, key, row = rng.randrange(B), rng.randrange(H), rng.randrange(S), rng.randrange(S)
            u = (torch.randint(0, 2, (128,), device=DEV, generator=gen).float() * 2 - 1).to(dtype)
            a = rng.choice([1.2, 3.0, 30.0])
            k[b_, key, h_] = a * u
            q[b_, row, h_] = a * u
        causal = "causal" in mode
        run = (lambda: flash_attention.forward(cfg, q, k, v)) if mode == "plain" else (lambda: flash_attention.forward_ex(cfg, q, k, v, causal=causal))
        out, again = run(), run()
        ok = torch.equal(out, again) and bool(torch.isfinite(out.float()).all())
        ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
        step = max(1, (1 << 26) // (H * S * S))
        for b0 in range(0, B, step):
            ref = eager(q[b0:b0 + step], k[b0:b0 + step], v[b0:b0 + step], causal)
            ok = ok and bool(((out[b0:b0 + step].float() - ref).abs() <= ulp * (1 + ref.abs())).all())
        n += 1
        if not ok:
            bad += 1
            print("FAIL", str(dtype), cfg.speculative_softmax, mode, B, H, S, "items per workgroup", B * H * ((S + 255) // 256) / 256, flush=True)
    print(f"soak (many items): {n} launches x 2 in {time.time() - t0:.0f} s, {bad} failures")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
