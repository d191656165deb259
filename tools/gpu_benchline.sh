#!/bin/bash
# The driver's command with bench.py as committed, and rocprofv3 --kernel-trace --stats of the same command (side legs off, so
# the stats cover the default kernel's launches only) on the SAME lease; then the other BASELINE configs.
# -> profiles/rNN/bench_c1.json + rocprof_kernel_stats.csv (+ bench_c2 / c3 / c4).  Usage: bash tools/gpu_benchline.sh [tag]
export PYTHONPATH=$PWD:$PYTHONPATH
TAG=${1:-r04}
OUT=gpurun_out/${TAG}_benchline
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
echo "== the driver's command"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c1.json 2> $OUT/bench.err; cut -c1-250 $OUT/bench_c1.json; tail -2 $OUT/bench.err
echo "== rocprofv3 --kernel-trace --stats of it (no side legs)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o fa -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-traffic --no-mfma-roof --no-cpu-baseline --hermetic-reps 0 --no-variants > $OUT/rocprof_bench.json 2> $OUT/rocprof.err; tail -1 $OUT/rocprof.err
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -3 $f; cp $f $OUT/rocprof_kernel_stats.csv; done
echo "== c2"; timeout 900 python bench.py --workload c2 --steps 20 --warmup 5 > $OUT/bench_c2.json 2>/dev/null; cut -c1-200 $OUT/bench_c2.json
for W in c3 c4; do echo "== $W"; timeout 900 python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --hermetic-reps 0 --no-mfma-roof > $OUT/bench_$W.json 2>/dev/null; cut -c1-200 $OUT/bench_$W.json; done
echo "== done"
