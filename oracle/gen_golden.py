#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference's Python oracles.

Runs only in the build container (needs /root/reference; the GPU box has no copy).
Nothing from the reference is copied: the reference modules are imported in place,
with their CUDA-only dependencies (flash_attn_2_cuda, flash_attn_3_cuda,
flash_attention, flash_attn, wurlitzer) stubbed in sys.modules, and only DATA
(seeded inputs + the outputs the reference's functions return) is written.

What each fixture pins:
  eager_<dtype>_<case>.npz   q,k,v (uint16 bit patterns), o_b16 = py_flash_attention(
                              upcast=False), o_f32 = py_flash_attention(upcast=True)
                              -- utils.py:137-162
  block_<case>.npz           O_final of tools/debug/debug.py:block_flash_attention
                              (rows of "warp 2", :50-57) on fp32 copies of bf16 inputs
  seam_<dtype>.npz           a case big enough that the persistent walk crosses item seams on a 256-CU chip
                              (3 x 1024 x 24 heads = 288 items of 256 rows) with planted logit spikes in a FIRST and in
                              a SECOND item of a workgroup (the speculative softmax's second pass): inputs by recipe
                              (seed + the spikes, rebuilt by tests/conftest.py:build_seam_inputs), outputs of
                              py_flash_attention as a ROW SAMPLE (every row of the spiked Q blocks + every 64th row)
  configs.json               get_kernels_to_build / progression short forms, FLOP model
                              values -- kernel_configs.py
Usage:  python oracle/gen_golden.py   (from the repo root)
"""
import io
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

CASES = {
    # name: (batch, seq, heads, d_head, seed)
    "a": (1, 256, 1, 128, 11),
    "b": (2, 256, 2, 128, 12),   # heads != 16: catches a hard-coded head stride (gotcha G1)
    "c": (1, 512, 1, 128, 13),
}


def import_reference():
    for name in ("flash_attn_2_cuda", "flash_attn_3_cuda", "flash_attention", "wurlitzer"):
        sys.modules.setdefault(name, types.ModuleType(name))
    fa = types.ModuleType("flash_attn")
    fa.flash_attn_func = None
    sys.modules.setdefault("flash_attn", fa)
    sys.modules["wurlitzer"].pipes = None
    sys.path.insert(0, os.path.join(REF, "py"))
    sys.path.insert(0, os.path.join(REF, "tools", "debug"))
    import flash_helpers.kernel_configs as kc
    import flash_helpers.test.utils as ut
    import debug as dbg

    return kc, ut, dbg


def u16(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


SEAM = dict(shape=(3, 1024, 24, 128), seed=17,
            # (batch, head, key, first q row, q rows, amplitude, sign seed): q rows and the key get amplitude * (+-1 vector)
            # item numbering of the persistent walk (n_bh = 72, 4 Q blocks): item 5 = (b 0, h 5, Q block 0), ordinal 0;
            # item 260 = (b 2, h 20, Q block 0), the SECOND item of workgroup 4; the third spike is mild (bf16 stays in
            # the first pass, fp16 does not)
            spikes=[(0, 5, 3, 40, 8, 30.0, 1), (2, 20, 700, 100, 4, 30.0, 2), (1, 7, 70, 600, 6, 1.107, 3)])


def build_seam_inputs(dtype):
    """The seam case's q, k, v from its recipe (the same function lives in tests/conftest.py: the GPU box has no copy
    of this file's reference imports)."""
    B, S, H, D = SEAM["shape"]
    gen = torch.Generator().manual_seed(SEAM["seed"])
    q, k, v = (torch.randn((B, S, H, D), generator=gen).to(dtype) for _ in range(3))
    for (b, h, key, row0, nrows, amp, sseed) in SEAM["spikes"]:
        g2 = torch.Generator().manual_seed(1000 + sseed)
        u = (torch.randint(0, 2, (D,), generator=g2).float() * 2 - 1) * amp
        k[b, key, h] = u.to(dtype)
        q[b, row0:row0 + nrows, h] = u.to(dtype)
    return q, k, v


def seam_row_sample():
    """(batch, row, head) index arrays of the stored sample: the spiked Q blocks whole, and every 64th row of everything."""
    B, S, H, D = SEAM["shape"]
    idx = set()
    for (b, h, key, row0, nrows, amp, sseed) in SEAM["spikes"]:
        blk = (row0 // 256) * 256
        idx.update((b, r, h) for r in range(blk, blk + 256))
    idx.update((b, r, h) for b in range(B) for r in range(5, S, 64) for h in range(H))
    idx = sorted(idx)
    return tuple(np.array([t[i] for t in idx], dtype=np.int64) for i in range(3))


def main():
    kc, ut, dbg = import_reference()
    os.makedirs(OUT, exist_ok=True)
    bi, ri, hi = seam_row_sample()
    for dtype, tag in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        q, k, v = build_seam_inputs(dtype)
        o_b16 = ut.py_flash_attention(q, k, v, upcast=False)
        o_f32 = ut.py_flash_attention(q, k, v, upcast=True)
        np.savez_compressed(
            os.path.join(OUT, f"seam_{tag}.npz"),
            shape=np.array(SEAM["shape"]), seed=np.array(SEAM["seed"]), spikes=np.array(SEAM["spikes"], dtype=np.float64),
            b=bi.astype(np.int16), r=ri.astype(np.int16), h=hi.astype(np.int16),
            o_b16=u16(o_b16[bi, ri, hi]), o_f32=u16(o_f32[bi, ri, hi]),
            # a checksum of the rebuilt inputs, so a different torch RNG cannot silently change the case
            qkv_sum=np.array([float(t.double().sum()) for t in (q, k, v)]),   # (float64: independent of the summation order)
        )
    for dtype, tag in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        for name, (B, S, H, D, seed) in CASES.items():
            torch.manual_seed(seed)
            q = torch.randn((B, S, H, D), dtype=dtype)
            k = torch.randn_like(q)
            v = torch.randn_like(q)
            o_b16 = ut.py_flash_attention(q, k, v, upcast=False)
            o_f32 = ut.py_flash_attention(q, k, v, upcast=True)
            np.savez(
                os.path.join(OUT, f"eager_{tag}_{name}.npz"),
                q=u16(q), k=u16(k), v=u16(v), o_b16=u16(o_b16), o_f32=u16(o_f32),
                shape=np.array([B, S, H, D]), seed=np.array(seed),
            )
    # blockwise trace: fp32 arithmetic on bf16-representable inputs, one head
    for name, (B_r, B_c, n_warps) in {"a": (128, 64, 4), "c": (64, 32, 4)}.items():
        B, S, H, D, seed = CASES[name]
        torch.manual_seed(seed)
        q = torch.randn((B, S, H, D), dtype=torch.bfloat16)
        k = torch.randn_like(q)
        v = torch.randn_like(q)
        q2, k2, v2 = (t[0, :, 0].float() for t in (q, k, v))
        o_final = dbg.block_flash_attention(D, q2, k2, v2, B_r, B_c, io.StringIO(), n_warps)
        rows = B_r // n_warps
        np.savez(
            os.path.join(OUT, f"block_{name}.npz"),
            o_final=o_final.numpy(), B_r=np.array(B_r), B_c=np.array(B_c),
            row_start=np.array(2 * rows), row_stop=np.array(3 * rows),
        )
    cfg = {
        "kernels_to_build": [c.short_form() for c in kc.get_kernels_to_build()],
        "kernels_to_build_cpp": [c.to_cpp_struct() for c in kc.get_kernels_to_build()],
        "autotune": [c.short_form() for c in kc.get_autotuning_kernel_configs()],
        "progression": [c.short_form() for c in kc.get_kernel_progression_configs()],
        "progression_all": [c.short_form() for c in kc.get_kernel_progression_configs(True)],
        "self_attn_flop_4_16_4096_128": kc.calc_self_attn_flop(4, 16, 4096, 128),
        "total_flop_4_16_4096_128_64_128": kc.calc_total_flop(4, 16, 4096, 128, 64, 128),
        "arithmetic_intensity_128_64_4096_128": kc.arithmetic_intensity(128, 64, 4096, 128),
        "smem_bytes_128_64": (128 + 2 * 64) * 128 * 2,
        "batch_size_for_seq_len": {str(k): v for k, v in ut.BATCH_SIZE_FOR_SEQ_LEN.items()},
        "benchmark_n_heads": ut.BENCHMARK_N_HEADS,
        "typed_name_example": {
            "name": "void flash_forward_kernel<FlashForwardKernelConfig{(c10::ScalarType)5, (int)128, (int)64, (int)64, (int)4, (bool)1, (bool)1, (bool)1, (int)0, (int)2, (int)0, (bool)1, (bool)1}>(FAForwardArgs)",
            "short": kc.parse_kernel_name_into_config(
                "void flash_forward_kernel<FlashForwardKernelConfig{(c10::ScalarType)5, (int)128, (int)64, (int)64, (int)4, (bool)1, (bool)1, (bool)1, (int)0, (int)2, (int)0, (bool)1, (bool)1}>(FAForwardArgs)"
            ).short_form(),
        },
    }
    with open(os.path.join(OUT, "configs.json"), "w") as f:
        json.dump(cfg, f, indent=1)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
