import torch, sys
sys.path.insert(0, '.')
import flash_attention_from_scratch_amd  # noqa
from flash_attention_from_scratch_amd import flash_attention
from flash_attention_from_scratch_amd.flash_helpers import kernel_configs as kc
def t(cfg, q, k, v, causal, n=20):
    for _ in range(5): flash_attention.forward_ex(cfg, q, k, v, causal=causal)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): flash_attention.forward_ex(cfg, q, k, v, causal=causal)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
P = kc.FlashForwardKernelConfig(kc.DType.BF16, 128, 256, 64, 4, True, True, True, 0, 0, 0, True, False)
A = kc.FlashForwardKernelConfig(kc.DType.BF16, 128, 128, 64, 4, True, True, True, 0, 0, 0, True, False)
Bc = kc.FlashForwardKernelConfig(kc.DType.BF16, 128, 256, 128, 8, True, True, True, 0, 0, 0, False, True)
for (B, S, H) in ((16, 300, 16), (16, 1000, 16), (8, 2500, 16), (4, 4000, 16), (2, 8191, 16), (4, 4096, 16)):
    q, k, v = (torch.randn((B, S, H, 128), dtype=torch.bfloat16, device='cuda') for _ in range(3))
    for causal in (False, True):
        fl = 4.0 * B * H * S * S * 128 * (0.5 if causal else 1.0)
        r = [fl / (t(c, q, k, v, causal) * 1e-3) / 1e12 for c in (P, A, Bc)]
        print(f"B={B} S={S} causal={int(causal)}: persistent {r[0]:7.1f}  (128,64,4) {r[1]:7.1f}  (256,128,8) {r[2]:7.1f} useful TFLOP/s")
