#!/bin/bash
# The round's closing lease: the files of profiles/<round>/ that belong to the driver's line, on the library as committed --
# GPU suite, smoke, the driver's bench command, rocprofv3 --kernel-trace --stats of the same command (same lease: VERDICT r02
# task 3), the interleaved speculative / lazy / pre-scaled comparison, the C2 sweep, the non-Gaussian data table, the soaks.
# Usage (repo root, GPU box): bash tools/gpu_final.sh [tag]
TAG=${1:-r03_final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
SPEC="(BF16, 128, 256, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer+spec_softmax"
LAZY="(BF16, 128, 256, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer"
echo "== pytest"; timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.txt 2>&1; tail -2 $OUT/pytest_gpu.txt
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
echo "== bench c1 (the driver's command)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c1.json 2> $OUT/bench.err; cut -c1-240 $OUT/bench_c1.json; tail -2 $OUT/bench.err
echo "== rocprof --kernel-trace --stats of the bench command (same lease)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o fa -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof > $OUT/rocprof_bench.json 2> $OUT/rocprof.err; tail -1 $OUT/rocprof.err
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -4 $f | cut -c1-200; cp $f $OUT/rocprof_kernel_stats.csv; done; rm -rf $OUT/prof
echo "== speculative / lazy / pre-scaled Q, interleaved, 200 steps each, two rounds"
for round in 1 2; do for K in "$SPEC" "$LAZY" "$SPEC+prescaled_q"; do
  timeout 600 python bench.py --steps 200 --warmup 5 --kernel "$K" --no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof > $OUT/b.json 2>>$OUT/bench.err
  python -c "import json;r=json.load(open('$OUT/b.json'));print('%8.1f TFLOP/s  %s MHz  %s' % (r['value'], r['clocks'].get('sclk_mhz',{}).get('mean'), r['config']['kernel']))" | tee -a $OUT/interleaved_c1.txt
done; done
echo "== c2 sweep"; timeout 900 python bench.py --workload c2 --steps 20 --warmup 5 > $OUT/bench_c2.json 2>>$OUT/bench.err; cut -c1-200 $OUT/bench_c2.json
echo "== non-Gaussian data"
: > $OUT/sink_data.txt
for D in randn sink heavy; do for T in bf16 fp16; do for K in spec lazy; do
  KK="$SPEC"; [ $K = lazy ] && KK="$LAZY"; [ $T = fp16 ] && KK="${KK/BF16/FP16}"
  timeout 600 python bench.py --steps 20 --warmup 5 --data $D --dtype $T --kernel "$KK" --no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof > $OUT/b.json 2>/dev/null
  python -c "import json;r=json.load(open('$OUT/b.json'));s=r['speculative'];print('%-6s %-5s %-12s %8.1f TFLOP/s   items %d redone %d (%.1f %%)' % ('$D','$T',r['config']['softmax_mode'],r['value'],s['items'],s['items_redone'],100*s['second_pass_fraction']))" | tee -a $OUT/sink_data.txt
done; done; done
rm -f $OUT/b.json
echo "== soaks"; timeout 400 python tools/soak.py 120 5 all > $OUT/soak.txt 2>&1; tail -2 $OUT/soak.txt
timeout 300 python tools/soak_many_items.py 60 > $OUT/soak_many_items.txt 2>&1; tail -2 $OUT/soak_many_items.txt
echo "== done"
