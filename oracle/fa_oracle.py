"""ctypes front-end of oracle/libfa_oracle.so plus torch restatements.

TEST INFRASTRUCTURE ONLY (see oracle/fa_oracle.c header).  Parity status: pinned
against tests/golden/*.npz, which oracle/gen_golden.py produced by importing the
reference's own Python oracles in the build container.

Functions
---------
eager_attention(q, k, v, upcast)      restates py_flash_attention
                                       (/root/reference/py/flash_helpers/test/utils.py:137-162)
blockwise_attention_torch(...)        restates block_flash_attention
                                       (/root/reference/tools/debug/debug.py:40-153), all rows
blockwise_forward(...)                C restatement of the device algorithm
                                       (forward_kernel.cuh:19-204, softmax.cuh)
eager_forward_c(...)                  C restatement of the eager arithmetic
sdpa_cpu(q, k, v)                     torch CPU scaled_dot_product_attention -- the
                                       timing baseline BASELINE.md section 3 names
"""

import ctypes
import math
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfa_oracle.so")
_lib = None

FP16, BF16 = 5, 15


def build(force=False):
    """gcc-compile fa_oracle.c (called by __graft_entry__.build() and on first use)."""
    src = os.path.join(_HERE, "fa_oracle.c")
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    ):
        subprocess.run(["make", "-C", _HERE, "-B", "libfa_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        u16p = ctypes.c_void_p
        i64 = ctypes.c_int64
        L.fa_oracle_forward_blockwise.restype = ctypes.c_int
        L.fa_oracle_forward_blockwise.argtypes = [
            u16p, u16p, u16p, u16p, ctypes.c_int, i64, i64, i64, i64, i64, i64, i64,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
        ]
        L.fa_oracle_forward_blockwise_masked.restype = ctypes.c_int
        L.fa_oracle_forward_blockwise_masked.argtypes = [
            u16p, u16p, u16p, u16p, ctypes.c_int, i64, i64, i64, i64, i64, i64, i64,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ]
        L.fa_oracle_forward_blockwise_lazy.restype = ctypes.c_int
        L.fa_oracle_forward_blockwise_lazy.argtypes = [
            u16p, u16p, u16p, u16p, ctypes.c_int, i64, i64, i64, i64, i64, i64, i64,
            ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
        ]
        L.fa_oracle_forward_blockwise_lazy_psq.restype = ctypes.c_int
        L.fa_oracle_forward_blockwise_lazy_psq.argtypes = L.fa_oracle_forward_blockwise_lazy.argtypes
        L.fa_oracle_forward_blockwise_spec.restype = ctypes.c_int
        L.fa_oracle_forward_blockwise_spec.argtypes = [
            u16p, u16p, u16p, u16p, ctypes.c_int, i64, i64, i64, i64, i64, i64, i64,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ]
        L.fa_oracle_forward_eager.restype = ctypes.c_int
        L.fa_oracle_forward_eager.argtypes = [
            u16p, u16p, u16p, u16p, ctypes.c_void_p, ctypes.c_int, i64, i64, i64, i64,
            i64, i64, i64, ctypes.c_int, ctypes.c_int,
        ]
        L.fa_oracle_b16_to_f32.restype = ctypes.c_float
        L.fa_oracle_b16_to_f32.argtypes = [ctypes.c_uint16, ctypes.c_int]
        L.fa_oracle_f32_to_b16.restype = ctypes.c_uint16
        L.fa_oracle_f32_to_b16.argtypes = [ctypes.c_float, ctypes.c_int]
        L.fa_oracle_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _code(dtype):
    if dtype == torch.float16:
        return FP16
    if dtype == torch.bfloat16:
        return BF16
    raise ValueError("only fp16 and bf16 are supported")


def _check(q, k, v):
    for t in (q, k, v):
        if t.device.type != "cpu" or not t.is_contiguous():
            raise ValueError("oracle wants contiguous CPU tensors")
    if not (q.shape == k.shape == v.shape) or q.dim() != 4:
        raise ValueError("q, k, v must share one (batch, seq, heads, d_head) shape")


def blockwise_forward(q, k, v, B_r, B_c, round_p=True, optimized_softmax=False,
                      return_stats=False, n_threads=0):
    """Device-algorithm restatement on (batch, seq, heads, d_head) 16-bit CPU tensors."""
    _check(q, k, v)
    B, S, H, D = q.shape
    o = torch.empty_like(q)
    m = torch.empty((B, H, S), dtype=torch.float32) if return_stats else None
    l = torch.empty((B, H, S), dtype=torch.float32) if return_stats else None
    rc = lib().fa_oracle_forward_blockwise(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), _code(q.dtype),
        B, S, H, D, q.stride(0), q.stride(1), q.stride(2), B_r, B_c,
        int(round_p), int(optimized_softmax),
        m.data_ptr() if return_stats else None, l.data_ptr() if return_stats else None,
        n_threads,
    )
    if rc != 0:
        raise RuntimeError(f"fa_oracle_forward_blockwise failed: {rc}")
    return (o, m, l) if return_stats else o


def blockwise_forward_lazy(q, k, v, B_r, B_c, tau=8.0, n_threads=0, prescaled_q=False):
    """Lazy-rescale restatement (NOT the reference's arithmetic: the MI355X 64-rows-per-wave
    variant's): O and l stay relative to a reference max that moves only when a row's max rose by
    more than `tau` in the base-2 exponent somewhere in its 32-row group.
    prescaled_q: the device's pre-scaled-Q option -- Q * c rounded to 16 bit once, the
    exponent is then the raw dot product."""
    _check(q, k, v)
    B, S, H, D = q.shape
    o = torch.empty_like(q)
    fn = lib().fa_oracle_forward_blockwise_lazy_psq if prescaled_q else lib().fa_oracle_forward_blockwise_lazy
    rc = fn(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), _code(q.dtype),
        B, S, H, D, q.stride(0), q.stride(1), q.stride(2), B_r, B_c, float(tau), n_threads,
    )
    if rc != 0:
        raise RuntimeError(f"fa_oracle_forward_blockwise_lazy failed: {rc}")
    return o


SPEC_TAU = 1e30  # "never move the reference max": the speculative schedule's first pass


def blockwise_forward_spec(q, k, v, B_r, B_c, kv_forward=True, n_threads=0, prescaled_q=False, alt_group=0):
    """The speculative first pass (NOT the reference's arithmetic): a row's reference is the row max of
    the first K / V block visited and never moves.  kv_forward: blocks first-to-last, as the persistent kernel's plain
    forms walk them since round 6 (False: last-to-first = blockwise_forward_lazy with tau = SPEC_TAU).  alt_group = G >= 2:
    the Q blocks with (qb // G) odd visit [block 0, then last-to-second] -- the device's alternating form for long
    sequences (kernel_configs.kv_walk_alternates gives G for a launch)."""
    _check(q, k, v)
    B, S, H, D = q.shape
    o = torch.empty_like(q)
    rc = lib().fa_oracle_forward_blockwise_spec(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), _code(q.dtype),
        B, S, H, D, q.stride(0), q.stride(1), q.stride(2), B_r, B_c,
        int(alt_group) if (kv_forward and alt_group >= 2) else int(bool(kv_forward)), int(bool(prescaled_q)), n_threads,
    )
    if rc != 0:
        raise RuntimeError(f"fa_oracle_forward_blockwise_spec failed: {rc}")
    return o


def blockwise_for_config(cfg, q, k, v, n_threads=0, masked=False, num_cus=256):
    """The CPU restatement of the arithmetic the device variant behind `cfg` performs on inputs that
    do not trip the speculative schedule's overflow check (those rows are redone with tau = 8).
    `masked`: the config's causal / ragged form (only the persistent kernel's is speculative)."""
    from flash_helpers import kernel_configs as kc

    psq = bool(getattr(cfg, "prescaled_q", False))
    if kc.uses_speculative_softmax(cfg, masked):
        return blockwise_forward_spec(q, k, v, min(cfg.B_r, q.shape[1]), cfg.B_c,
                                      kv_forward=kc.walks_kv_forward(cfg, masked, q.shape[1]), n_threads=n_threads,
                                      prescaled_q=psq,
                                      alt_group=kc.kv_walk_alternates(cfg, q.shape[0] * q.shape[2], q.shape[1], masked, num_cus))
    if kc.uses_lazy_rescale(cfg, q.shape[1]):
        return blockwise_forward_lazy(q, k, v, cfg.B_r, cfg.B_c, n_threads=n_threads, prescaled_q=psq)
    return blockwise_forward(q, k, v, cfg.B_r, cfg.B_c, optimized_softmax=cfg.optimized_softmax,
                             n_threads=n_threads)


def blockwise_forward_masked(q, k, v, B_r, B_c, causal=False, optimized_softmax=False, n_threads=0):
    """Widened modes (not in the reference): any seq_len, optional causal mask."""
    _check(q, k, v)
    B, S, H, D = q.shape
    o = torch.empty_like(q)
    rc = lib().fa_oracle_forward_blockwise_masked(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), _code(q.dtype),
        B, S, H, D, q.stride(0), q.stride(1), q.stride(2), B_r, B_c,
        int(optimized_softmax), int(causal), n_threads,
    )
    if rc != 0:
        raise RuntimeError(f"fa_oracle_forward_blockwise_masked failed: {rc}")
    return o


def eager_attention_masked(q, k, v, causal=False, upcast=True):
    """softmax(mask(q k^T / sqrt(d))) v in fp32 (or the input dtype): the eager statement of
    the widened modes, same op order as eager_attention."""
    dtype_in = q.dtype
    if upcast:
        q, k, v = q.float(), k.float(), v.float()
    scores = torch.einsum("bqhd,bkhd->bqhk", q, k) / (q.shape[-1] ** 0.5)
    if causal:
        S = q.shape[1]
        dead = torch.ones((S, S), dtype=torch.bool, device=q.device).triu(1)
        scores = scores.masked_fill(dead[None, :, None, :], float("-inf"))
    out = torch.einsum("bqhk,bkhd->bqhd", scores.softmax(dim=-1), v)
    return out.to(dtype_in) if upcast else out


def eager_forward_c(q, k, v, upcast=True, return_f32=False, n_threads=0):
    _check(q, k, v)
    B, S, H, D = q.shape
    o = torch.empty_like(q)
    f32 = torch.empty((B, S, H, D), dtype=torch.float32) if return_f32 else None
    rc = lib().fa_oracle_forward_eager(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(),
        f32.data_ptr() if return_f32 else None, _code(q.dtype),
        B, S, H, D, q.stride(0), q.stride(1), q.stride(2), int(upcast), n_threads,
    )
    if rc != 0:
        raise RuntimeError(f"fa_oracle_forward_eager failed: {rc}")
    return (o, f32) if return_f32 else o


def eager_attention(q, k, v, upcast=False):
    """softmax(q k^T / sqrt(d)) v with the reference's eager op order
    (utils.py:137-162): divide by d**0.5 after the matmul, softmax over keys,
    optional fp32 upcast with the result cast back."""
    dtype_in = q.dtype
    if upcast:
        q, k, v = q.float(), k.float(), v.float()
    scores = torch.einsum("bqhd,bkhd->bqhk", q, k) / (q.shape[-1] ** 0.5)
    probs = scores.softmax(dim=-1)
    out = torch.einsum("bqhk,bkhd->bqhd", probs, v)
    return out.to(dtype_in) if upcast else out


def blockwise_attention_torch(q2d, k2d, v2d, B_c, rows=None):
    """fp32 torch restatement of debug.py:block_flash_attention's fused branch
    (:122-137) for the rows `rows` (a slice) of one head: reverse KV order, raw
    running max, base-2 exponent with scale d^-0.5*log2(e), fp32 P."""
    Q = q2d[rows] if rows is not None else q2d
    d_head = q2d.shape[-1]
    scale = (d_head ** -0.5) * math.log2(math.e)
    ks, vs = k2d.split(B_c, dim=0), v2d.split(B_c, dim=0)
    M = torch.full((Q.shape[0], 1), float("-inf"), dtype=Q.dtype)
    L = torch.zeros_like(M)
    O = torch.zeros_like(Q)
    for i in reversed(range(len(ks))):
        S = Q @ ks[i].T
        M_new = torch.maximum(M, S.max(dim=-1, keepdim=True).values)
        resc = 2 ** ((M - M_new) * scale)
        P = 2 ** (S * scale - M_new * scale)
        L = L * resc + P.sum(dim=-1, keepdim=True)
        O = O * resc + P @ vs[i]
        M = M_new
    return O / L


def sdpa_cpu(q, k, v):
    """torch CPU SDPA on (B,S,H,d) tensors -> (B,S,H,d); the cpu_baseline kernel."""
    out = torch.nn.functional.scaled_dot_product_attention(
        q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    )
    return out.transpose(1, 2)


def tolerance_rule(out, ref_b16, ref_f32):
    """The reference's accuracy bar (py/flash_helpers/test/test.py:57-61):
    max|out - eager_b16| <= 2 * max|eager_b16 - eager_f32|.  Returns (lhs, rhs)."""
    lhs = (out.float() - ref_b16.float()).abs().max().item()
    rhs = 2 * (ref_b16.float() - ref_f32.float()).abs().max().item()
    return lhs, rhs


def as_u16(t):
    return t.view(torch.int16).numpy().view(np.uint16)


def from_u16(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(dtype)
