"""Host cost of one flash_attention.forward() call (checks + ctypes + hipLaunchKernel) on a tiny problem, and where it goes
(cProfile): 13-14 us per call on the GPU box -- a third of the shortest C2 launch (S = 512: 40 us), so the sweeps are not host-bound."""
import time, torch, sys
sys.path.insert(0,'.')
import flash_attention
from flash_helpers import kernel_configs as kc
for dt in (kc.DType.BF16, kc.DType.FP16):
    cfg = kc.best_config(dt)
    q,k,v = (torch.randn(1,256,1,128,device='cuda',dtype=cfg.dtype.to_torch_dtype()) for _ in range(3))
    o = torch.empty_like(q)
    for _ in range(100): flash_attention.forward(cfg,q,k,v,o)
    torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(5000): flash_attention.forward(cfg,q,k,v,o)
    t1=time.perf_counter()-t
    torch.cuda.synchronize()
    t2=time.perf_counter()-t
    print(dt.name, "host per call %.1f us, incl. drain %.1f us" % (t1/5000*1e6, t2/5000*1e6))
plain = kc.FlashForwardKernelConfig(kc.DType.BF16,128,256,64,4,True,True,True,0,0,0,True,False)
q,k,v = (torch.randn(1,256,1,128,device='cuda',dtype=torch.bfloat16) for _ in range(3)); o=torch.empty_like(q)
for _ in range(100): flash_attention.forward(plain,q,k,v,o)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(5000): flash_attention.forward(plain,q,k,v,o)
t1=time.perf_counter()-t; torch.cuda.synchronize()
print("plain config host per call %.1f us" % (t1/5000*1e6))
import cProfile, pstats
cfg = kc.best_config(kc.DType.BF16)
pr=cProfile.Profile(); pr.enable()
for _ in range(2000): flash_attention.forward(cfg,q,k,v,o)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
