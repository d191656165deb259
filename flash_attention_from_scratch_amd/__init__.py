"""MI355X-native Flash-Attention-2 forward behind the API of
sonnyli/flash_attention_from_scratch.

Sub-packages mirror the reference's two pip packages and its extension module:
    flash_attention_from_scratch_amd.flash_attention          (forward, forward_timed)
    flash_attention_from_scratch_amd.flash_attention_kernels  (forward(cfg,q,k,v,o,benchmark))
    flash_attention_from_scratch_amd.flash_helpers            (kernel_configs, test.utils)
The repo root also carries thin `flash_attention`, `flash_attention_kernels` and
`flash_helpers` aliases so code written against the reference imports unchanged.
"""

__version__ = "0.1.0"
