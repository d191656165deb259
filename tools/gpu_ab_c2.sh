#!/bin/bash
# interleaved A/B of two builds of the library (lib/libfa_old.so, lib/libfa_new.so) on the c2 sweep and c1
# (bench.py protocol, clocks preconditioned; no PMC passes).  Usage: bash tools/gpu_ab_c2.sh [reps]
L=flash_attention_from_scratch_amd/lib
mkdir -p gpurun_out/ab
for rep in $(seq 1 ${1:-3}); do for w in old new; do
  cp $L/libfa_$w.so $L/libfa_hip.so
  python bench.py --workload c2 --steps 40 --warmup 5 --precondition-ms 250 --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w c2', f\"{d['value']:7.1f}\", ' '.join(f\"{v['tflops']:7.1f}\" for v in d['per_seq_len'].values()))"
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w c1', f\"{d['value']:7.1f} median {d['roofline']['flop_per_launch']/d['roofline']['kernel_ms_per_launch']['median']/1e9:7.1f}\")"
done; done | tee gpurun_out/ab/ab_c2.txt
cp $L/libfa_new.so $L/libfa_hip.so
