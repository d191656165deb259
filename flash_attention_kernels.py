"""Drop-in alias of the reference's extension module name."""
from flash_attention_from_scratch_amd.flash_attention_kernels import forward  # noqa: F401
