#!/bin/bash
# round 3: the pre-scaled Q -- parity, error report, interleaved timing against the exact-c kernel
TAG=${1:-r03b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.txt 2>&1; tail -15 $OUT/pytest_gpu.txt
echo "== error report"; timeout 600 python tools/psq_error.py > $OUT/prescaled_q_error.txt 2>&1; cat $OUT/prescaled_q_error.txt
SPEC="(BF16, 128, 256, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer+spec_softmax"
for round in 1 2; do for K in "$SPEC" "$SPEC+prescaled_q" "(BF16, 128, 256, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer" "(BF16, 128, 256, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer+prescaled_q"; do
  timeout 600 python bench.py --steps 200 --warmup 5 --kernel "$K" --no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof > $OUT/b.json 2>>$OUT/bench.err; python -c "import json;r=json.load(open('$OUT/b.json'));print(round(r['value'],1),r['clocks'].get('sclk_mhz',{}).get('mean'),r['clocks'].get('power_w',{}).get('mean'),r['config']['kernel'])"
done; done
for K in "$SPEC" "$SPEC+prescaled_q"; do
  echo "== c2 $K"; timeout 900 python bench.py --workload c2 --steps 20 --warmup 5 --no-traffic --kernel "$K" > $OUT/c2.json 2>>$OUT/bench.err; python -c "import json;r=json.load(open('$OUT/c2.json'));print(r['value'],{k:round(v['tflops']) for k,v in r['per_seq_len'].items()})"
done
echo "== pipe counters of the pre-scaled kernel"; timeout 600 python bench.py --steps 20 --warmup 5 --kernel "$SPEC+prescaled_q" --no-cpu-baseline --hermetic-reps 0 --no-mfma-roof > $OUT/bench_c1_psq.json 2>>$OUT/bench.err; python -c "import json;r=json.load(open('$OUT/bench_c1_psq.json'));print(r['value'],r['roofline'].get('pipe_counters'))"
echo "== done"
