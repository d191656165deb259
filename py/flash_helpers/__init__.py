"""Drop-in alias: `flash_helpers.*` resolves to flash_attention_from_scratch_amd.flash_helpers.*"""
