#!/bin/bash
# interleaved A/B of two builds of the library (lib/libfa_old.so, lib/libfa_new.so) under the bench.py protocol,
# then the GPU tests on the new one.  Usage: bash tools/gpu_ab.sh [pytest -k expression]
export PYTHONPATH=$PWD:$PYTHONPATH
L=flash_attention_from_scratch_amd/lib
mkdir -p gpurun_out/ab
for rep in 1 2 3; do for w in old new; do
  cp $L/libfa_$w.so $L/libfa_hip.so
  python bench.py --steps 60 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', f\"{d['value']:8.1f} TF  {d['ms_per_step']:.4f} ms\")"
done; done | tee gpurun_out/ab/ab.txt
for w in old new; do
  cp $L/libfa_$w.so $L/libfa_hip.so
  KERNELS=best timeout 300 python flash_attention_from_scratch_amd/tools/pt_bench.py --seq_lens 512,1024,2048 --num_repeats 20 --num_warmups 5 --no-ref 2>/dev/null | cut -d, -f1-5,12 | grep -v "^Kernel" | sed "s/^/$w /"
done | tee gpurun_out/ab/ab_small.txt
cp $L/libfa_new.so $L/libfa_hip.so
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 ${1:+-k "$1"} > gpurun_out/ab/pytest.txt 2>&1; tail -4 gpurun_out/ab/pytest.txt
