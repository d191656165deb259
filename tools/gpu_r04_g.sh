#!/bin/bash
# round 4, call G: fp16 guard threshold of the speculative first pass, 2^11 (lib/libfa_old.so, rounds 3-4) vs 2^13 (libfa_new.so)
export PYTHONPATH=$PWD:$PYTHONPATH
L=flash_attention_from_scratch_amd/lib
OUT=gpurun_out/r04g
mkdir -p $OUT
: > $OUT/fp16_guard_threshold.txt
for rep in 1 2 3; do for w in old new; do
  cp $L/libfa_$w.so $L/libfa_hip.so
  for args in "--workload c3 --steps 10 --warmup 3" "--workload c1 --dtype fp16 --steps 40 --warmup 5"; do
    python bench.py $args --no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', '$args'.split()[1], '%.1f TFLOP/s  %.4f ms  sustained %s' % (d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('tflops')), d['speculative'])" | tee -a $OUT/fp16_guard_threshold.txt
  done
done; done
cp $L/libfa_new.so $L/libfa_hip.so
echo "== fp16 S=32768 (rows pass 2^13 on their own there), new"; python - <<'PY' | tee -a gpurun_out/r04g/fp16_guard_threshold.txt
import torch, time
import flash_attention, flash_attention_kernels
from flash_helpers import kernel_configs as kc
from dataclasses import replace
for S in (8192, 16384, 32768):
    cfg = replace(kc.best_config(kc.DType.FP16, S), adaptive_softmax=False)
    q, k, v = (torch.randn((1, S, 16, 128), dtype=torch.float16, device="cuda") for _ in range(3))
    stats = torch.zeros(2, dtype=torch.int32, device="cuda")
    out, _ = flash_attention_kernels.forward(cfg, q, k, v, None, stats=stats)
    lazy = flash_attention.forward(replace(cfg, speculative_softmax=False), q, k, v)
    torch.cuda.synchronize()
    print("S", S, "items/redone", stats.tolist(), "max |spec - lazy|", float((out.float() - lazy.float()).abs().max()))
PY
echo "== pytest (new)"; timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > $OUT/pytest_gpu.txt 2>&1; tail -6 $OUT/pytest_gpu.txt
echo "== done"
