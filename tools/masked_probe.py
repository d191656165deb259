#!/usr/bin/env python3
"""Why does the masked (causal / ragged) build of the persistent kernel trail the plain one with nothing masked?
Times the plain and the masked variant on the same shape (mask off, causal on), speculative / lazy / adaptive, and
prints fa_fwd_stats (items, items computed twice) and the adaptive state beside each: a second pass or a demotion
shows here, a slower visit body does not.  Usage: python tools/masked_probe.py [S B H]"""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), os.pardir)))
import flash_attention  # noqa: E402
from flash_attention_from_scratch_amd import _capi  # noqa: E402
from flash_helpers import kernel_configs as kc  # noqa: E402


def timed(fn, reps=30, warm_s=0.3):
    # clocks preconditioned like bench.py does (the governor ramps from idle over ~0.2 s; without it the FIRST row of the
    # table read 10 % low), no synchronize between the warm launches and the timed ones
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        for _ in range(16):
            fn()
        torch.cuda.current_stream().synchronize()
    for _ in range(8):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    S, B, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (4096, 4, 16)
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    q, k, v = (torch.randn((B, S, H, 128), generator=g).to(torch.bfloat16).to(dev) for _ in range(3))
    o = torch.empty_like(q)
    base = kc.best_config(kc.DType.BF16, S, masked=True)
    modes = {
        "speculative": dataclasses.replace(base, adaptive_softmax=False, speculative_softmax=True),
        "adaptive": base,
        "lazy": kc.as_native(base, speculative_softmax=False) if hasattr(kc, "as_native") else base,
    }
    print(f"shape B={B} S={S} H={H}  cfg {base.short_form()}")
    for name, cfg in modes.items():
        for label, kw in (("plain (forward)", None), ("masked, mask off", dict(causal=False)), ("masked, causal", dict(causal=True))):
            _capi.adaptive_reset(0)
            st = torch.zeros(2, dtype=torch.int32, device=dev)
            if kw is None:
                fn = lambda: flash_attention.forward(cfg, q, k, v, o)  # noqa: E731
                one = fn
            else:
                fn = lambda: flash_attention.forward_ex(cfg, q, k, v, o, **kw)  # noqa: E731
                one = lambda: flash_attention.forward_ex(cfg, q, k, v, o, stats=st, **kw)  # noqa: E731
            ms = timed(fn)
            one()
            torch.cuda.synchronize()
            fl = 4.0 * B * H * S * S * 128 * (0.5 if kw and kw.get("causal") else 1.0)
            print(f"{name:12s} {label:18s} {ms:8.4f} ms {fl / ms / 1e9:8.1f} TFLOP/s   items {int(st[0])} redone {int(st[1])}   adaptive {_capi.adaptive_state(0)}")


if __name__ == "__main__":
    main()
