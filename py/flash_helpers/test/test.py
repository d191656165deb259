"""Drop-in alias of the reference's test entry: `python py/flash_helpers/test/test.py` (or `python -m flash_helpers.test.test`)."""
import os
import sys

if __name__ == "__main__":  # run as a script: the repository root is not on sys.path yet (this file lives in
    # <root>/flash_helpers/test/ and, for the flash_helpers distribution, in <root>/py/flash_helpers/test/)
    _here = os.path.dirname(os.path.abspath(__file__))
    for _up in (2, 3):
        _root = os.path.abspath(os.path.join(_here, *([os.pardir] * _up)))
        if os.path.isdir(os.path.join(_root, "flash_attention_from_scratch_amd")):
            sys.path.insert(0, _root)
            break

from flash_attention_from_scratch_amd.flash_helpers.test.test import *  # noqa: E402,F401,F403
from flash_attention_from_scratch_amd.flash_helpers.test.test import (  # noqa: E402,F401
    FlashAttentionTestBF16,
    FlashAttentionTestFP16,
)

if __name__ == "__main__":
    import unittest

    unittest.main(verbosity=2)
