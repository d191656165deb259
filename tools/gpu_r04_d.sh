#!/bin/bash
# round 4, call D: (1) does the clock sampler's start-up in front of the first timed launch cost the driver's 20-step number?
# (2) rocprofv3 --stats of the bench command without the variant / robustness legs (the default kernel's launches only)
export PYTHONPATH=$PWD:$PYTHONPATH
OUT=gpurun_out/r04d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-traffic --no-mfma-roof --no-cpu-baseline --hermetic-reps 0 --no-variants"
: > $OUT/sampler_start.txt
for rep in 1 2 3 4; do for mode in early late; do
  E=0; [ $mode = late ] && E=1
  FA_BENCH_SAMPLER_LATE=$E $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_launch']
print('$mode value %.1f  kernel_ms mean %.4f median %.4f max %.4f first %.4f argmax %d  sclk %s' % (d['value'], k['mean'], k['median'], k['max'], k['first'], k['argmax'], d['clocks'].get('sclk_mhz',{}).get('mean')))" | tee -a $OUT/sampler_start.txt
done; done
echo "== rocprof --kernel-trace --stats of the driver's command without the side legs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o fa -- $B > $OUT/rocprof_bench.json 2> $OUT/rocprof.err; tail -2 $OUT/rocprof.err
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -4 $f; cp $f $OUT/rocprof_kernel_stats.csv; done
cut -c1-300 $OUT/rocprof_bench.json
echo "== the driver's command, full line"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c1.json 2> $OUT/bench.err; cut -c1-250 $OUT/bench_c1.json
echo "== c2"; timeout 900 python bench.py --workload c2 --steps 20 --warmup 5 --no-traffic > $OUT/bench_c2.json 2>/dev/null; cut -c1-200 $OUT/bench_c2.json
echo "== done"
