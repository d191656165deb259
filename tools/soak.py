#!/usr/bin/env python3
"""Soak of the persistent kernel: random (batch, heads, seq_len, causal, dtype, speculative) launches for a time budget,
each checked against fp32 attention computed by torch on the same device, launched twice (bits must repeat) and, every
few launches, with another stream hammering the memory system.  Usage: python tools/soak.py [seconds] [seed] [all | ring]   (all: every device variant of the library; ring: the one-Q-tile-per-wave forms of (128, 64, 4)+buffer)"""
import random
import sys
import time

import torch

sys.path.insert(0, ".")
import flash_attention  # noqa: E402
from flash_helpers import kernel_configs as kc  # noqa: E402

DEV = "cuda:0"


def eager(q, k, v, causal):
    qf, kf, vf = (t.float().permute(0, 2, 1, 3) for t in (q, k, v))
    s = qf @ kf.transpose(-1, -2) / (q.shape[-1] ** 0.5)
    if causal:
        S = q.shape[1]
        s = s.masked_fill(torch.ones(S, S, dtype=torch.bool, device=q.device).triu(1), float("-inf"))
    return (torch.softmax(s, dim=-1) @ vf).permute(0, 2, 1, 3)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    side = torch.cuda.Stream()
    noise = torch.empty(512 << 20, dtype=torch.int8, device=DEV)
    t0, n, bad = time.time(), 0, 0
    every = len(sys.argv) > 3 and sys.argv[3] == "all"   # every device variant of the library instead of the persistent kernel
    pool = kc.get_all_supported_configs() + kc.get_d64_kernel_configs() if every else []
    # "ring": only the reference's winning shape, (B_r 128, B_c 64, 4 warps) + buffer, plain and speculative, at multiples
    # of 256 keys -- the persistent kernel's one-Q-tile-per-wave forms (round 5), spiked every second launch
    ring = len(sys.argv) > 3 and sys.argv[3] == "ring"
    if ring:
        every = True
        pool = [kc.as_native(c, speculative_softmax=sp) for c in kc.get_kernels_to_build() if kc.has_ring_form(c) for sp in (False, True)]
        assert pool and all(kc.has_ring_form(c) for c in pool)
    while time.time() - t0 < budget:
        dtype, name = rng.choice(((torch.bfloat16, kc.DType.BF16), (torch.float16, kc.DType.FP16)))
        spec = rng.random() < 0.7
        cfg = kc.NativeKernelConfig(name, 128, 256, 64, 4, True, True, True, 0, 0, 0, True, False, speculative_softmax=spec)
        if every:
            cfg = rng.choice(pool)
            dtype, spec = cfg.dtype.to_torch_dtype(), kc.wants_speculative(cfg)
        masked = rng.random() < 0.5 and not ring
        S = rng.choice([64, 100, 200, 256, 300, 500, 512, 768, 1000, 1024, 1500, 2048, 3000, 4096]) if masked else 256 * rng.randint(1, 20)
        if every and not masked and not ring:
            S = max(cfg.B_r, cfg.B_c) * rng.randint(1, 24)
        causal = masked and rng.random() < 0.6
        B, H = rng.randint(1, 6), rng.choice([1, 2, 3, 5, 8, 16, 24])
        if not every and not masked and rng.random() < 0.04:
            # round 6: a long sequence whose launch takes the form that alternates the K / V direction by rounds
            # (batch * heads a multiple of 8, 64 Q blocks = two rounds of an XCD's workgroups)
            S, B, H = 16384, 1, 8
        while B * H * S * S > 3e9:
            B = max(1, B - 1)
            H = max(1, H // 2)
        gen = torch.Generator(device=DEV).manual_seed(rng.randrange(1 << 30))
        q, k, v = (torch.randn((B, S, H, cfg.d_head), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        if rng.random() < (0.5 if ring else 0.3):   # a spike: the speculative first pass fails somewhere
            b_, h_, key = rng.randrange(B), rng.randrange(H), rng.randrange(S)
            u = (torch.randint(0, 2, (cfg.d_head,), device=DEV, generator=gen).float() * 2 - 1).to(dtype)
            a = rng.choice([1.2, 3.0, 30.0])
            k[b_, key, h_] = a * u
            q[b_, rng.randrange(S), h_] = a * u
        if n % 3 == 0:
            with torch.cuda.stream(side):
                noise.zero_()
        run = (lambda: flash_attention.forward_ex(cfg, q, k, v, causal=causal)) if masked else (lambda: flash_attention.forward(cfg, q, k, v))
        try:
            out, again = run(), run()
        except RuntimeError as exc:   # (all: this configuration has no masked form in the library)
            if "no causal / ragged-length variant" not in str(exc) and "requested native options" not in str(exc):
                raise
            continue
        ref = eager(q, k, v, causal)
        ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
        if getattr(cfg, "prescaled_q", False):
            # the pre-scaled Q's logit error grows with |k|: a 30-sigma key moves a row's weight on it by ~1 %
            ulp *= 16.0
        ok = bool(torch.isfinite(out.float()).all()) and bool(((out.float() - ref).abs() <= ulp * (1 + ref.abs())).all()) and torch.equal(out, again)
        n += 1
        if not ok:
            bad += 1
            print("FAIL", cfg.short_form() if every else "", str(dtype), "spec" if spec else "lazy", "masked" if masked else "plain", "causal" if causal else "", B, H, S,
                  "max err", float((out.float() - ref).abs().max()), "repeat", torch.equal(out, again), flush=True)
    torch.cuda.synchronize()
    print(f"soak: {n} launches x 2 in {time.time() - t0:.0f} s, {bad} failures")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
