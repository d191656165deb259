"""`flash_attention` -- the reference's public API, unchanged
(/root/reference/flash_attention/__init__.py:7-17):

    forward(kernel_cfg, q, k, v, o=None) -> Tensor
    forward_timed(kernel_cfg, q, k, v, o=None) -> (Tensor, milliseconds)
"""

from .. import flash_attention_kernels


def forward(kernel_cfg, q, k, v, o=None):
    return flash_attention_kernels.forward(kernel_cfg, q, k, v, o, benchmark=False)[0]


def forward_timed(kernel_cfg, q, k, v, o=None):
    out, runtime_ms = flash_attention_kernels.forward(kernel_cfg, q, k, v, o, benchmark=True)
    return out, runtime_ms


def forward_ex(kernel_cfg, q, k, v, o=None, causal=False, timed=False, stats=None):
    """Scope wideners beyond the reference API: optional causal mask, and any
    seq_len (not only multiples of B_r / B_c).  `stats`: optional device tensor of two 32-bit counters
    (items computed, items the speculative softmax computed twice; fa_fwd_stats in include/fa_hip.h).
    Returns Tensor, or (Tensor, ms) if timed."""
    out, ms = flash_attention_kernels.forward(kernel_cfg, q, k, v, o, benchmark=timed, causal=causal,
                                              allow_ragged=True, stats=stats)
    return (out, ms) if timed else out
