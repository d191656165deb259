#!/bin/bash
TAG=${1:-r03c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
SPEC="(BF16, 128, 256, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer+spec_softmax"
LAZY="(BF16, 128, 256, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer"
for round in 1 2; do for K in "$SPEC" "$LAZY" "$SPEC+prescaled_q"; do
  timeout 600 python bench.py --steps 200 --warmup 5 --kernel "$K" --no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof > $OUT/b.json 2>>$OUT/bench.err; python -c "import json;r=json.load(open('$OUT/b.json'));print(round(r['value'],1),r['clocks'].get('sclk_mhz',{}).get('mean'),r['config']['kernel'])"
done; done
: > $OUT/sink_data.txt
for D in randn sink heavy; do for T in bf16 fp16; do for K in spec lazy; do
  KK="$SPEC"; [ $K = lazy ] && KK="$LAZY"; [ $T = fp16 ] && KK="${KK/BF16/FP16}"
  timeout 600 python bench.py --steps 20 --warmup 5 --data $D --dtype $T --kernel "$KK" --no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof > $OUT/b.json 2>/dev/null
  python -c "import json;r=json.load(open('$OUT/b.json'));s=r['speculative'];print('%-6s %-5s %-12s %8.1f TFLOP/s   items %d redone %d (%.1f %%)' % ('$D','$T',r['config']['softmax_mode'],r['value'],s['items'],s['items_redone'],100*s['second_pass_fraction']))" | tee -a $OUT/sink_data.txt
done; done; done
timeout 900 python bench.py --workload c2 --steps 20 --warmup 5 --no-traffic > $OUT/c2.json 2>>$OUT/bench.err; python -c "import json;r=json.load(open('$OUT/c2.json'));print(r['value'],{k:round(v['tflops']) for k,v in r['per_seq_len'].items()})"
timeout 300 python tools/soak.py 40 11 2>&1 | tail -2
