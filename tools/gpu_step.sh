#!/bin/bash
# One short gpurun call while iterating on the kernel: GPU tests, the tune64 A/B against the previous round's binary,
# the item / seam traces, a quick bench line.  Usage: bash tools/gpu_step.sh <tag> [skip-tests]
TAG=${1:-step}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
L=flash_attention_from_scratch_amd/lib
if [ -z "$2" ]; then
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
fi
echo "== tune64: this tree, the round-4 binary, this tree again (same lease)"
timeout 300 $L/tune64 reps=6 > $OUT/tune64_a.txt 2>&1; timeout 300 $L/tune64 reps=6 > $OUT/tune64_b.txt 2>&1
grep -h "S=" $OUT/tune64_a.txt | cut -c1-150; echo "-- again"; grep -h "S=" $OUT/tune64_b.txt | cut -c1-150
echo "== trace64_items / seam"; for a in "512 16 16" "1024 16 16" "4096 4 16"; do timeout 120 $L/trace64_items $a 2>&1 | grep -E "^==|mean over" | tail -2; done > $OUT/trace64_items.txt; timeout 120 $L/trace64_seam 512 16 16 2>&1 | grep -E "first seam" | tail -1 >> $OUT/trace64_items.txt; timeout 120 $L/trace64_seam 4096 4 16 2>&1 | grep -E "first seam" | tail -1 >> $OUT/trace64_items.txt; cut -c1-400 $OUT/trace64_items.txt
echo "== masked probe"; timeout 300 python tools/masked_probe.py 4096 4 16 > $OUT/masked_probe.txt 2>&1; cat $OUT/masked_probe.txt | tail -12
echo "== bench c1 (the driver command, full)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c1.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench_c1.json; tail -2 $OUT/bench.err
echo "== bench c2"; timeout 600 python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c2.json 2>> $OUT/bench.err; cut -c1-300 $OUT/bench_c2.json
echo "== done"
