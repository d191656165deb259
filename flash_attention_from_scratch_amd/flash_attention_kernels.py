"""`flash_attention_kernels` -- Python mirror of the reference's pybind module
(/root/reference/src/flash_attention.cu:34-140) over the C ABI of libfa_hip.so.

    forward(kernel_cfg, q, k, v, o, benchmark=False) -> (Tensor, float)

Same argument meaning, same checks in the same order with the same messages
(RuntimeError, as TORCH_CHECK raises), same ownership rule for `o`, launch on
torch's current stream, blocking only when benchmark=True.  PyTorch-ROCm is
plumbing here (device memory, current stream); the computation is the HIP kernel
behind fa_fwd_launch.  There is no eager/CPU fallback: a missing library or a
non-gfx950 device raises.

`kernel_cfg.optimized_softmax` has the reference's meaning only (the first KV block skips the
rescale; same result with or without).  This build's extensions are asked for explicitly: a
`flash_helpers.kernel_configs.NativeKernelConfig` (what `best_config()` returns) carries
`speculative_softmax` / `prescaled_q`, which travel in `fa_fwd_opts` beside the 13-field key.  A plain
13-field config with `optimized_softmax` selects the speculative softmax only under
FA_ALLOW_SPECULATIVE=1 (the round-2 mapping, kept for sweeps).
"""

import ctypes
import os

import torch

from . import _capi


_STATS_DTYPES = tuple(d for d in (torch.int32, getattr(torch, "uint32", None)) if d is not None)  # (torch.uint32: torch >= 2.3)


def _check_input(t, name):
    # CHECK_INPUT, src/include/cuda_utils.cuh:5-11
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def _native_options(kernel_cfg):
    """-> (speculative, prescaled_q); speculative is False, True, "adaptive" (fa_speculative_mode) or None = "where the
    config has a speculative variant" (FA_ALLOW_SPECULATIVE=1 on a plain config)."""
    spec = bool(getattr(kernel_cfg, "speculative_softmax", False))
    if spec and getattr(kernel_cfg, "adaptive_softmax", False):
        spec = "adaptive"
    if not spec and getattr(kernel_cfg, "optimized_softmax", False) and os.environ.get("FA_ALLOW_SPECULATIVE", "") == "1":
        spec = None  # "if the config has a speculative variant": resolved against the registry below
    return spec, bool(getattr(kernel_cfg, "prescaled_q", False))


def forward(kernel_cfg, q, k, v, o=None, benchmark=False, causal=False, allow_ragged=False, stats=None):
    """Reference signature plus keyword-only-by-convention wideners (default off, so the
    reference behaviour -- including its errors for seq_len not a multiple of the tiles -- is
    unchanged): `causal` applies a causal mask, `allow_ragged` accepts any seq_len, `stats` is an
    int32 / uint32 device tensor of >= 2 elements whose [0] the kernel increases by the number of work
    items it computed and whose [1] by the number the speculative softmax had to compute twice
    (fa_fwd_stats)."""
    masked = bool(causal or allow_ragged)
    _check_input(q, "q")
    _check_input(k, "k")
    _check_input(v, "v")

    q_dtype = q.dtype
    if q_dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("Only fp16 and bf16 are supported")
    if k.dtype != q_dtype or v.dtype != q_dtype:
        raise RuntimeError("Input tensors must have the same data type")
    if q.dim() != 4:
        raise RuntimeError("q must have shape (batch, seq_len, n_heads, d_head)")

    cfg = _capi.make_config(kernel_cfg)
    lib = _capi.load()
    if not lib.fa_fwd_supported(ctypes.byref(cfg)):
        raise RuntimeError("Kernel configuration was not found in flash_kernels (libfa_hip.so registry)")
    if masked and not lib.fa_fwd_masked_supported(ctypes.byref(cfg)):
        raise RuntimeError("Kernel configuration has no causal / ragged-length variant in libfa_hip.so")
    speculative, prescaled_q = _native_options(kernel_cfg)
    if speculative is None:  # FA_ALLOW_SPECULATIVE=1: where the variant exists
        probe = _capi.make_opts(causal=causal, allow_ragged=allow_ragged, speculative=True, prescaled_q=prescaled_q)
        speculative = bool(lib.fa_fwd_ex_supported(ctypes.byref(cfg), ctypes.byref(probe)))
    if speculative and masked:
        # the masked forms of the 32- / 16-rows-per-wave kernels keep the running max (no speculative build): the request is
        # dropped there, as kc.softmax_mode(cfg, masked=True) and the oracle's blockwise_for_config report
        probe = _capi.make_opts(causal=causal, allow_ragged=allow_ragged, speculative=True, prescaled_q=prescaled_q)
        if not lib.fa_fwd_ex_supported(ctypes.byref(cfg), ctypes.byref(probe)):
            plain = _capi.make_opts(causal=causal, allow_ragged=allow_ragged, speculative=False, prescaled_q=prescaled_q)
            if lib.fa_fwd_ex_supported(ctypes.byref(cfg), ctypes.byref(plain)):
                speculative = False
    if speculative or prescaled_q:
        probe = _capi.make_opts(causal=causal, allow_ragged=allow_ragged, speculative=speculative, prescaled_q=prescaled_q)
        if not lib.fa_fwd_ex_supported(ctypes.byref(cfg), ctypes.byref(probe)):
            raise RuntimeError("Kernel configuration has no device variant with the requested native options "
                               f"(speculative_softmax={speculative}, prescaled_q={prescaled_q}, masked={masked}) in libfa_hip.so")
    cfg_dtype = kernel_cfg.dtype.to_torch_dtype() if hasattr(kernel_cfg.dtype, "to_torch_dtype") else None
    if cfg_dtype is None:
        cfg_dtype = {5: torch.float16, 15: torch.bfloat16}[int(kernel_cfg.dtype)]
    if cfg_dtype != q_dtype:
        raise RuntimeError("Kernel configuration dtype does not match input dtype")

    if q.shape != k.shape:
        raise RuntimeError("Query and key tensors have same shape")
    if q.shape != v.shape:
        raise RuntimeError("Query and value tensors have same shape")
    batch, seq_len, n_heads, d_head = q.shape
    if not allow_ragged and seq_len % cfg.B_r != 0:
        raise RuntimeError("Only multiples of B_r are supported for seq_len Q currently")
    if not allow_ragged and seq_len % cfg.B_c != 0:
        raise RuntimeError("Only multiples of B_c are supported for seq_len K currently")

    if o is not None:
        if o.dtype != q_dtype:
            raise RuntimeError("Output tensor must have the same dtype as inputs")
        if o.shape != q.shape:
            raise RuntimeError("Query and output tensors have same shape")
        _check_input(o, "o")
    else:
        o = torch.empty_like(q)

    args = _capi.FaFwdArgs(
        q=q.data_ptr(), k=k.data_ptr(), v=v.data_ptr(), o=o.data_ptr(),
        batch=batch, seq_len=seq_len, n_heads=n_heads, d_head=d_head,
        batch_stride=q.stride(0), seq_stride=q.stride(1), head_stride=q.stride(2),
        cfg=cfg,
    )
    stats_ptr = None
    if stats is not None:
        if not stats.is_cuda or stats.device != q.device or stats.dtype not in _STATS_DTYPES \
                or stats.numel() < 2 or not stats.is_contiguous():
            raise RuntimeError("stats must be a contiguous int32 / uint32 tensor of >= 2 elements on q's device")
        stats_ptr = stats.data_ptr()
    with torch.cuda.device(q.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream)
        if masked or speculative or prescaled_q or stats_ptr is not None:
            ms = ctypes.c_float(0.0)
            opts = _capi.make_opts(causal=causal, allow_ragged=allow_ragged, speculative=speculative,
                                   prescaled_q=prescaled_q, ms=ms if benchmark else None, stats_ptr=stats_ptr)
            _capi.check(lib.fa_fwd_launch_ex(ctypes.byref(args), ctypes.byref(opts), stream))
            return o, float(ms.value)
        if benchmark:
            ms = ctypes.c_float(0.0)
            _capi.check(lib.fa_fwd_launch_timed(ctypes.byref(args), stream, ctypes.byref(ms)))
            return o, float(ms.value)
        _capi.check(lib.fa_fwd_launch(ctypes.byref(args), stream))
    return o, 0.0
