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


def build_seam_inputs(dtype, z):
    """q, k, v of tests/golden/seam_<tag>.npz from its recipe (oracle/gen_golden.py:build_seam_inputs restated: seeded
    N(0, 1) plus planted logit spikes); checked against the checksums the generator stored."""
    B, S, H, D = (int(x) for x in z["shape"])
    gen = torch.Generator().manual_seed(int(z["seed"]))
    q, k, v = (torch.randn((B, S, H, D), generator=gen).to(dtype) for _ in range(3))
    for (b, h, key, row0, nrows, amp, sseed) in z["spikes"]:
        g2 = torch.Generator().manual_seed(1000 + int(sseed))
        u = (torch.randint(0, 2, (D,), generator=g2).float() * 2 - 1) * float(amp)
        k[int(b), int(key), int(h)] = u.to(dtype)
        q[int(b), int(row0):int(row0) + int(nrows), int(h)] = u.to(dtype)
    sums = [float(t.double().sum()) for t in (q, k, v)]
    assert np.allclose(sums, z["qkv_sum"], rtol=0, atol=1e-6), (sums, z["qkv_sum"])
    return q, k, v


def load_seam_golden(tag):
    """-> dict: q, k, v (rebuilt), b / r / h index tensors of the stored row sample, o_b16 / o_f32 rows (n, d_head)."""
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[tag]
    z = np.load(os.path.join(GOLDEN, f"seam_{tag}.npz"))
    q, k, v = build_seam_inputs(dtype, z)
    out = {"q": q, "k": k, "v": v, "dtype": dtype, "spikes": z["spikes"]}
    for name in ("b", "r", "h"):
        out[name] = torch.from_numpy(z[name].astype(np.int64))
    for name in ("o_b16", "o_f32"):
        out[name] = torch.from_numpy(np.ascontiguousarray(z[name]).view(np.int16)).view(dtype)
    return out


@pytest.fixture(scope="session")
def golden_loader():
    return load_eager_golden
