"""Launch the default kernel at C1 on random inputs (3 warm-up + 3 launches), then on zero inputs (same),
for a rocprofv3 --pmc pass around it: tools/gpu_pmc_zero.sh compares wave cycles and durations of the two."""
import sys
import torch
sys.path.insert(0, ".")
import flash_attention_from_scratch_amd  # noqa: F401
from flash_attention_from_scratch_amd import flash_attention
from flash_attention_from_scratch_amd.flash_helpers import kernel_configs as kc

cfg = kc.best_config(kc.DType.BF16, 4096)
q, k, v = (torch.randn((4, 4096, 16, 128), dtype=torch.bfloat16, device="cuda") for _ in range(3))
o = torch.empty_like(q)
for data in ("random", "zeros"):
    if data == "zeros":
        q.zero_(); k.zero_(); v.zero_()
    for _ in range(30):
        flash_attention.forward(cfg, q, k, v, o)
    torch.cuda.synchronize()
