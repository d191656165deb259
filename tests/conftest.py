import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_eager_golden(tag, case):
    """-> dict of torch tensors q,k,v,o_b16,o_f32 (CPU) for tests/golden/eager_<tag>_<case>.npz"""
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[tag]
    z = np.load(os.path.join(GOLDEN, f"eager_{tag}_{case}.npz"))
    shape = tuple(int(x) for x in z["shape"])
    out = {}
    for name in ("q", "k", "v", "o_b16", "o_f32"):
        arr = np.ascontiguousarray(z[name]).view(np.int16)
        out[name] = torch.from_numpy(arr).view(dtype).reshape(shape)
    out["dtype"] = dtype
    return out


@pytest.fixture(scope="session")
def golden_loader():
    return load_eager_golden
