#!/bin/bash
TAG=${1:-s}
OUT=gpurun_out/$TAG
mkdir -p $OUT

for K in "256,64" "256,128" "128,64"; do
  KERNELS=$K timeout 900 python flash_attention_from_scratch_amd/tools/pt_bench.py --seq_lens 512,1024,2048,4096,8192,16384 --num_repeats 16 --num_warmups 4 --no-ref > $OUT/seq_${K/,/_}.csv 2> $OUT/seq.err
done
tail -2 $OUT/seq.err
