#!/bin/bash
# sustained (bench.py protocol) comparison of candidate configs, interleaved twice
export PYTHONPATH=$PWD:$PYTHONPATH
K1="(BF16, 128, 256, 128, 8): async+eager+swizzled+load_0_0_0_tiles"
K2="(BF16, 128, 256, 64, 8): async+eager+swizzled+load_0_0_0_tiles+buffer"
K3="(BF16, 128, 128, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer"
K4="(BF16, 128, 256, 128, 8): async+eager+swizzled+load_0_0_0_tiles+opt_softmax"
for rep in 1 2; do for K in "$K1" "$K2" "$K3" "$K4"; do
  python bench.py --steps ${STEPS:-100} --warmup 20 --no-cpu-baseline --kernel "$K" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(f\"{d['value']:8.1f} TF  {d['ms_per_step']:.4f} ms  {d['config']['kernel']}\")"
done; done
