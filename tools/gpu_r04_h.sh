#!/bin/bash
# round 4, call H: GPU tier on the fp16 guard threshold 2^13 + the C3 line with traffic and pipe counters
export PYTHONPATH=$PWD:$PYTHONPATH
OUT=gpurun_out/r04h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
echo "== pytest"; timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > $OUT/pytest_gpu.txt 2>&1; tail -6 $OUT/pytest_gpu.txt
echo "== c3"; timeout 900 python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --hermetic-reps 0 --no-mfma-roof > $OUT/bench_c3.json 2>/dev/null; cut -c1-220 $OUT/bench_c3.json
echo "== staircase probe (fp16): which steps pass"; python - <<'PY'
import torch
import flash_attention_kernels
from flash_helpers import kernel_configs as kc
for step in (0.1, 0.2, 0.3, 0.4, 0.5):
    B, H, S = 2, 4, 4096
    gen = torch.Generator(device="cuda").manual_seed(5)
    q, k, v = (torch.randn((B, S, H, 128), dtype=torch.float16, device="cuda", generator=gen) for _ in range(3))
    a = (step / 0.12751743) ** 0.5
    t = (S - 1 - torch.arange(S, device="cuda")) // 64
    q[..., 0] = a
    k[..., 0] = (a * t.float()).view(1, S, 1).to(torch.float16)
    cfg = kc.NativeKernelConfig(kc.DType.FP16, 128, 256, 64, 4, True, True, True, 0, 0, 0, True, False, speculative_softmax=True)
    stats = torch.zeros(2, dtype=torch.int32, device="cuda")
    flash_attention_kernels.forward(cfg, q, k, v, None, stats=stats)
    print("step", step, "items/redone", stats.tolist())
PY
echo "== done"
