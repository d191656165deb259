"""Drop-in alias: `import flash_attention` resolves to the MI355X build."""
from flash_attention_from_scratch_amd.flash_attention import forward, forward_ex, forward_timed  # noqa: F401
