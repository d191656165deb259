#!/usr/bin/env python3
"""Seeded launches of the persistent kernel through whichever build of the library FA_HIP_LIB names (default: the
product, lib/libfa_hip.so), one line per case: a SHA-256 of the output and whether three launches agreed bit for bit.

    python tools/jitter_check.py                                       # product library
    FA_HIP_LIB=flash_attention_from_scratch_amd/lib/libfa_hip_jitter.so python tools/jitter_check.py

tests/test_gpu_parity.py::test_jitter_build_matches_the_product_bit_for_bit runs both and compares the lines: the
timing-perturbed build (FA_JITTER, csrc/fa_fwd_kernel64.hpp) sleeps pseudo-randomly in front of every DMA piece, sync
point and operand wait, so its waves meet every step of the ring protocol in another order -- and must still produce
the product's bits.  The cases cross item seams (more items than workgroups), take the second pass of the speculative
softmax (a planted spike), the causal and ragged forms, both dtypes, and the lazy (running max) schedule."""
import hashlib
import sys

import torch

sys.path.insert(0, ".")
import flash_attention  # noqa: E402
from flash_helpers import kernel_configs as kc  # noqa: E402

DEV = "cuda:0"
# (dtype, batch, seq_len, heads, speculative, causal, ragged, spike)
CASES = [
    ("bf16", 3, 1024, 24, True, False, False, False),    # 288 items: seams on every workgroup
    ("bf16", 3, 1024, 24, True, False, False, True),     # ... and a second pass
    ("bf16", 3, 1024, 24, False, False, False, True),    # lazy schedule, a rescale
    ("fp16", 2, 2048, 16, True, False, False, False),
    ("fp16", 2, 2048, 16, False, False, False, True),
    ("bf16", 2, 2048, 16, True, True, False, False),     # causal form
    ("bf16", 5, 1000, 7, True, False, True, False),      # ragged form
    ("fp16", 5, 1000, 7, True, True, True, True),        # ragged + causal + a spike
    ("bf16", 16, 512, 16, True, False, False, False),    # 2 items of 8 visits per workgroup
    ("bf16", 1, 8192, 8, True, False, False, False),     # 128 visits per item
    ("fp16", 1, 16384, 8, True, False, False, False),    # round 6: the long-sequence form (every second round walks K / V [0, last .. 1])
    ("bf16", 1, 16384, 8, True, False, False, True),     # ... and a second pass behind it (the spiked row's item walks that way: Q block 32)
]


def main():
    from flash_attention_from_scratch_amd import _capi

    print("# library:", _capi.LIB_PATH, _capi.version())
    for i, (dt, B, S, H, spec, causal, ragged, spike) in enumerate(CASES):
        dtype, name = (torch.bfloat16, kc.DType.BF16) if dt == "bf16" else (torch.float16, kc.DType.FP16)
        cfg = kc.NativeKernelConfig(name, 128, 256, 64, 4, True, True, True, 0, 0, 0, True, False, speculative_softmax=spec)
        gen = torch.Generator(device=DEV).manual_seed(1234 + i)
        q, k, v = (torch.randn((B, S, H, 128), dtype=dtype, device=DEV, generator=gen) for _ in range(3))
        if spike:
            u = (torch.randint(0, 2, (128,), device=DEV, generator=gen).float() * 2 - 1).to(dtype)
            k[B - 1, S // 3, H - 1] = 3.0 * u
            q[B - 1, S // 2, H - 1] = 3.0 * u
        masked = causal or ragged
        run = (lambda: flash_attention.forward_ex(cfg, q, k, v, causal=causal)) if masked else (lambda: flash_attention.forward(cfg, q, k, v))
        outs = [run() for _ in range(3)]
        torch.cuda.synchronize()
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        h = hashlib.sha256(outs[0].cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:24]
        print(f"case {i} {dt} B{B} S{S} H{H} spec={int(spec)} causal={int(causal)} ragged={int(ragged)} spike={int(spike)} "
              f"finite={int(bool(torch.isfinite(outs[0].float()).all()))} repeat={int(same)} sha={h}")


if __name__ == "__main__":
    main()
