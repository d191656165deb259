"""Test / bench fixtures with the reference's public names
(/root/reference/py/flash_helpers/test/utils.py).

The reference compares against Dao-AILab's flash_attn_2_cuda / flash_attn_3_cuda
wheels (utils.py:20-97); those are CUDA-only, so the comparator here is torch's
scaled_dot_product_attention on the same device (reference_forward_kernel_v2 /
_v3 keep their names and call signatures).
"""

import time
from dataclasses import dataclass

import torch

# batch per seq_len keeps tokens ~constant in the sweeps (utils.py:9-16)
BATCH_SIZE_FOR_SEQ_LEN = {
    512: 16,
    1024: 16,
    2048: 16,
    4096: 16,
    8192: 8,
    16384: 4,
}
BENCHMARK_N_HEADS = 16


@dataclass(frozen=True)
class QKVConfig:
    n_heads: int
    d_head: int

    batch_size: int
    seq_len: int

    dtype: torch.dtype
    device: torch.device

    @property
    def shape(self):
        return (self.batch_size, self.seq_len, self.n_heads, self.d_head)


def generate_qkv(cfg: QKVConfig, seed=None):
    """N(0,1) q, k, v of layout (batch, seq, heads, d_head) (utils.py:112-121).
    The reference is unseeded; pass `seed` for reproducible fixtures."""
    gen = None
    if seed is not None:
        gen = torch.Generator(device=cfg.device)
        gen.manual_seed(seed)
    q, k, v = (
        torch.randn(cfg.shape, dtype=cfg.dtype, device=cfg.device, generator=gen)
        for _ in range(3)
    )
    return q, k, v


def generate_qkvo(cfg: QKVConfig, seed=None):
    """q, k, v, o carved from ONE allocation in the order q, o, k, v (utils.py:124-134)."""
    gen = None
    if seed is not None:
        gen = torch.Generator(device=cfg.device)
        gen.manual_seed(seed)
    slab = torch.empty((4,) + cfg.shape, dtype=cfg.dtype, device=cfg.device)
    q, o, k, v = slab[0], slab[1], slab[2], slab[3]
    for t in (q, k, v):
        t.normal_(generator=gen)
    return q, k, v, o


def py_flash_attention(q, k, v, upcast: bool = False):
    """Eager oracle: softmax(q k^T / sqrt(d)) v on (batch, seq, heads, d) tensors,
    optionally in fp32 with the result cast back (utils.py:137-162)."""
    dtype_in = q.dtype
    if upcast:
        q, k, v = q.float(), k.float(), v.float()
    scores = torch.einsum("bqhd,bkhd->bqhk", q, k) / (q.shape[-1] ** 0.5)
    out = torch.einsum("bqhk,bkhd->bqhd", scores.softmax(dim=-1), v)
    return out.to(dtype_in) if upcast else out


def sdpa_attention(q, k, v):
    """torch SDPA on the (batch, seq, heads, d) layout."""
    out = torch.nn.functional.scaled_dot_product_attention(
        q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    )
    return out.transpose(1, 2).contiguous()


def reference_forward_kernel_v2(q, k, v, o=None):
    out = sdpa_attention(q, k, v)
    if o is not None:
        o.copy_(out)
        return o
    return out


def reference_forward_kernel_v3(q, k, v, o=None):
    return reference_forward_kernel_v2(q, k, v, o)


def reference_forward_kernel_v2_timed(q, k, v, o=None):
    """(out, milliseconds) with device events around the comparator."""
    if q.is_cuda:
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        out = reference_forward_kernel_v2(q, k, v, o)
        stop.record()
        stop.synchronize()
        return out, start.elapsed_time(stop)
    t0 = time.perf_counter()
    out = reference_forward_kernel_v2(q, k, v, o)
    return out, (time.perf_counter() - t0) * 1e3


def error_stats(expected, actual, atol=1e-5, rtol=1e-3):
    """(#mismatched, % mismatched, max |diff|) with isclose(atol, rtol) (utils.py:165-174)."""
    close = torch.isclose(expected, actual, atol=atol, rtol=rtol)
    mismatched = close.numel() - close.sum()
    return mismatched, 100 * mismatched / expected.numel(), (expected - actual).abs().max()


def evaluate_kernel(cfg, out_ref, out):
    mismatched, percent, max_diff = error_stats(out_ref, out)
    print(f"{cfg.short_form()}")
    print(f"  Mismatched elements: {mismatched} / {out.numel()} ({percent:.1f}%)")
    print(f"  Greatest absolute difference: {max_diff}")


def tolerance_check(out, ref_b16, ref_f32):
    """The reference's accuracy bar (test.py:57-61): returns (lhs, rhs) with the
    test passing when lhs <= rhs, lhs = max|out - ref_b16|, rhs = 2 max|ref_b16 - ref_f32|."""
    lhs = (out - ref_b16).abs().max().item()
    rhs = 2 * (ref_b16 - ref_f32).abs().max().item()
    return lhs, rhs


def get_cuda_device_info(device_idx=0):
    """Name / arch / memory / CU count (utils.py:187-197; 'cuda' is torch's name for HIP)."""
    if not torch.cuda.is_available():
        raise RuntimeError("CUDA not available")
    dev = torch.cuda.get_device_properties(device_idx)
    return {
        "name": dev.name,
        "compute_capability": getattr(dev, "gcnArchName", f"{dev.major}.{dev.minor}"),
        "total_memory": f"{dev.total_memory / (1024**3):.2f} GB",
        "multi_processor_count": dev.multi_processor_count,
    }


def is_mi355x():
    info = get_cuda_device_info()
    return "gfx950" in str(info["compute_capability"]) or "MI355" in info["name"]


def is_a100():
    return False
