#!/bin/bash
# How long does the clock governor need?  The driver's command (20 timed steps) behind 0 / 400 / 1000 / 2000 / 4000 ms of untimed
# launches, fresh process each, two rounds on one box; and 2000 timed steps for the sustained rate.
OUT=gpurun_out/${1:-precondition}; mkdir -p $OUT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof"
for round in 1 2; do for ms in 0 400 1000 2000 4000; do
  timeout 300 $B --precondition-ms $ms > $OUT/b.json 2>/dev/null
  python -c "import json;r=json.load(open('$OUT/b.json'));print('precondition %5d ms: %7.1f TFLOP/s  %s MHz %s W' % ($ms, r['value'], r['clocks'].get('sclk_mhz',{}).get('mean'), r['clocks'].get('power_w',{}).get('mean')))" | tee -a $OUT/precondition_sweep.txt
done; done
timeout 300 python bench.py --steps 2000 --warmup 5 --no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof > $OUT/b.json 2>/dev/null
python -c "import json;r=json.load(open('$OUT/b.json'));print('2000 timed steps      : %7.1f TFLOP/s  %s MHz %s W' % (r['value'], r['clocks'].get('sclk_mhz',{}).get('mean'), r['clocks'].get('power_w',{}).get('mean')))" | tee -a $OUT/precondition_sweep.txt
rm -f $OUT/b.json
