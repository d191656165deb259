from flash_attention_from_scratch_amd.flash_helpers.test.utils import *  # noqa: F401,F403
