#!/bin/bash
# round 4, call E: (1) GPU tier on the refactored adaptive policy; (2) bench.py preconditioning: continuous (committed) vs a
# synchronize after every 8 launches (rounds 2-3), alternating, the driver's command without the side legs
export PYTHONPATH=$PWD:$PYTHONPATH
OUT=gpurun_out/r04e
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-traffic --no-mfma-roof --no-cpu-baseline --hermetic-reps 0 --no-variants"
: > $OUT/precondition_ab.txt
for rep in 1 2 3 4 5; do for mode in continuous sync8; do
  E=0; [ $mode = sync8 ] && E=1
  FA_BENCH_PRECONDITION_SYNC=$E $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_launch']
print('$mode value %.1f  kernel_ms mean %.4f median %.4f max %.4f first %.4f  sustained %.1f  sclk %s power %s  pre_steps %d' % (d['value'], k['mean'], k['median'], k['max'], k['first'], d['sustained']['tflops'], d['clocks'].get('sclk_mhz',{}).get('mean'), d['clocks'].get('power_w',{}).get('mean'), d['precondition']['untimed_steps']))" | tee -a $OUT/precondition_ab.txt
done; done
echo "== c2, both"; for E in 0 1; do FA_BENCH_PRECONDITION_SYNC=$E timeout 600 python bench.py --workload c2 --steps 20 --warmup 5 --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sync8' if $E else 'continuous', round(d['value'],1), {k:round(v['tflops']) for k,v in d['per_seq_len'].items()})" | tee -a $OUT/precondition_ab.txt; done
echo "== done"
