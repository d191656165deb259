"""Drop-in alias of the reference's test entry: `python flash_helpers/test/test.py`."""
import os
import sys

if __name__ == "__main__":  # run as a script: the repository root is not on sys.path yet
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from flash_attention_from_scratch_amd.flash_helpers.test.test import *  # noqa: E402,F401,F403
from flash_attention_from_scratch_amd.flash_helpers.test.test import (  # noqa: E402,F401
    FlashAttentionTestBF16,
    FlashAttentionTestFP16,
)

if __name__ == "__main__":
    import unittest

    unittest.main(verbosity=2)
