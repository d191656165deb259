from flash_attention_from_scratch_amd.flash_helpers.kernel_configs import *  # noqa: F401,F403
from flash_attention_from_scratch_amd.flash_helpers.kernel_configs import (  # noqa: F401
    _parse_flash_forward_demanged_name,
    _parse_flash_forward_demanged_name_with_types,
    _parse_short_form_flash_forward_kernel_config,
)
