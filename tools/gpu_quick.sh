#!/bin/bash
# quick iteration: subset of parity tests + native sweep at the headline shape
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "${PYTEST_K:-golden or reference_suite or rescale or determinism}" > $OUT/pytest_gpu.txt 2>&1; tail -6 $OUT/pytest_gpu.txt
echo "== sweep"; KERNELS=${KERNELS:-native} timeout 900 python flash_attention_from_scratch_amd/tools/pt_bench.py --seq_lens ${SEQ:-4096} --batch ${BATCH:-4} --num_repeats 20 --num_warmups 5 --no-ref ${SWEEP_ARGS} > $OUT/sweep.csv 2> $OUT/sweep.err; cut -d, -f1-5,12-13 $OUT/sweep.csv | head -${HEAD:-30}; tail -3 $OUT/sweep.err
