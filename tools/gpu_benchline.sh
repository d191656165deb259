#!/bin/bash
# The driver's bench command and `rocprofv3 --kernel-trace --stats` of the same command on ONE lease (task 3):
# profiles/<round>/bench_c1.json and rocprof_kernel_stats.csv belong to each other.  Usage: bash tools/gpu_benchline.sh
OUT=gpurun_out/r03_final2; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c1.json 2> $OUT/bench.err; cut -c1-200 $OUT/bench_c1.json; tail -2 $OUT/bench.err
python -c "import json;r=json.load(open('$OUT/bench_c1.json'));print(json.dumps(r['variants'])[:600]); print(r['roofline']['frac'], r['roofline'].get('frac_of_mfma_only_random'), r['protocols']['hermetic']['tflops'])"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o fa -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof > $OUT/rocprof_bench.json 2> $OUT/rocprof.err; tail -1 $OUT/rocprof.err
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -3 $f | cut -c1-200; cp $f $OUT/rocprof_kernel_stats.csv; done; rm -rf $OUT/prof
