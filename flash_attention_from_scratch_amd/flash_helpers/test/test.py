"""unittest entry of the flash_helpers test suite -- `python py/flash_helpers/test/test.py` or
`python -m flash_helpers.test.test` -- the counterpart of the reference's
py/flash_helpers/test/test.py:17-99: one test per config of get_kernels_to_build(), on the
reference's fixture (seq_len 2048, batch BATCH_SIZE_FOR_SEQ_LEN[2048], BENCHMARK_N_HEADS heads,
d_head 128, N(0,1) inputs on cuda:0), with the reference's bar

    max|kernel - eager_16bit|  <=  2 * max|eager_16bit - eager_fp32|        (test.py:51-61)

Classes FlashAttentionTestFP16 / FlashAttentionTestBF16 carry test methods named
test_fp16_<i>_<config> / test_bf16_<i>_<config> (what `parameterized.expand` would generate there;
that package is not a dependency here: the methods are made in a loop).  Unlike the reference the
inputs are seeded, so a failure reproduces.  KERNELS=native (or any get_kernel_configs key) widens
the list beyond the reference's built set.
"""
import os
import re
import unittest

import torch

import flash_attention
from flash_helpers.kernel_configs import DType, get_kernel_configs, get_kernels_to_build
from flash_helpers.test.utils import (
    BATCH_SIZE_FOR_SEQ_LEN,
    BENCHMARK_N_HEADS,
    QKVConfig,
    generate_qkv,
    py_flash_attention,
)

SEQ_LEN = 2048
DEVICE = "cuda:0"


def configs_under_test():
    key = os.environ.get("KERNELS", "")
    return get_kernel_configs(key) if key else get_kernels_to_build()


class _Fixture:
    """Per-dtype inputs and the two eager results, made once per class."""

    torch_dtype = None

    @classmethod
    def setUpClass(cls):
        if not torch.cuda.is_available():
            raise unittest.SkipTest("the flash_attention tests need an MI355X (no CPU route exists)")
        cls.inputs, cls.eager_b16, cls.eager_f32 = {}, {}, {}
        for d_head in (128,):
            shape = QKVConfig(n_heads=BENCHMARK_N_HEADS, d_head=d_head, batch_size=BATCH_SIZE_FOR_SEQ_LEN[SEQ_LEN],
                              seq_len=SEQ_LEN, dtype=cls.torch_dtype, device=torch.device(DEVICE))
            q, k, v = generate_qkv(shape, seed=2048)
            cls.inputs[d_head] = (q, k, v)
            cls.eager_b16[d_head] = py_flash_attention(q, k, v, upcast=False)
            cls.eager_f32[d_head] = py_flash_attention(q, k, v, upcast=True)

    def check_config(self, cfg):
        q, k, v = self.inputs[cfg.d_head]
        out = flash_attention.forward(cfg, q, k, v)
        lhs = (out - self.eager_b16[cfg.d_head]).abs().max().item()
        rhs = (self.eager_b16[cfg.d_head] - self.eager_f32[cfg.d_head]).abs().max().item()
        self.assertTrue(torch.isfinite(out.float()).all().item(), str(cfg))
        self.assertLessEqual(lhs, 2 * rhs, str(cfg))


class FlashAttentionTestFP16(_Fixture, unittest.TestCase):
    torch_dtype = torch.float16


class FlashAttentionTestBF16(_Fixture, unittest.TestCase):
    torch_dtype = torch.bfloat16


def _attach(cls, prefix, dtype):
    for i, cfg in enumerate(c for c in configs_under_test() if c.dtype == dtype and c.d_head == 128):
        def method(self, cfg=cfg):
            self.check_config(cfg)

        method.__doc__ = str(cfg)
        setattr(cls, f"{prefix}_{i}_" + re.sub(r"\W+", "_", str(cfg)).strip("_"), method)


_attach(FlashAttentionTestFP16, "test_fp16", DType.FP16)
_attach(FlashAttentionTestBF16, "test_bf16", DType.BF16)


if __name__ == "__main__":
    unittest.main(verbosity=2)
