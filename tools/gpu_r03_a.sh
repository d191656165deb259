#!/bin/bash
# round 3, first GPU call: parity of the new ABI + the measurements the VERDICT asks for on non-Gaussian data
TAG=${1:-r03a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $OUT/pytest_gpu.txt 2>&1; tail -15 $OUT/pytest_gpu.txt
echo "== rocprofv3 -L (counter names)"; timeout 120 rocprofv3 -L > $OUT/counters.txt 2>&1; grep -i -E "DRAM|MALL|EA0_RDREQ|EA0_WRREQ|TCC_REQ|TCC_READ" $OUT/counters.txt | cut -c1-160 | head -40
echo "== bench c1 (driver's command)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c1.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench_c1.json; tail -2 $OUT/bench.err
python - <<PY
import json
r=json.load(open("$OUT/bench_c1.json"))
print({k:r.get(k) for k in ("value","speculative","protocols")}); print(r["roofline"].get("mfma_only"), r["roofline"].get("frac_of_mfma_only_random"))
PY
for D in sink heavy; do for T in bf16 fp16; do
  echo "== c1 --data $D --dtype $T (default kernel)"; timeout 600 python bench.py --steps 20 --warmup 5 --data $D --dtype $T --no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof > $OUT/bench_c1_${D}_${T}.json 2>>$OUT/bench.err; python -c "import json;r=json.load(open('$OUT/bench_c1_${D}_${T}.json'));print(r['value'],r['config']['kernel'],r['speculative'])"
  K=$(python -c "from flash_helpers import kernel_configs as kc; from dataclasses import replace; c=kc.best_config(kc.DType.${T^^}); print(replace(c, speculative_softmax=not c.speculative_softmax).short_form())")
  echo "== same, other softmax mode: $K"; timeout 600 python bench.py --steps 20 --warmup 5 --data $D --dtype $T --kernel "$K" --no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof > $OUT/bench_c1_${D}_${T}_other.json 2>>$OUT/bench.err; python -c "import json;r=json.load(open('$OUT/bench_c1_${D}_${T}_other.json'));print(r['value'],r['config']['kernel'],r['speculative'])"
done; done
echo "== c1 fp16 randn, both modes, with pipe counters (why fp16 trails bf16)"
timeout 600 python bench.py --steps 20 --warmup 5 --dtype fp16 --no-cpu-baseline --hermetic-reps 0 > $OUT/bench_c1_fp16_lazy.json 2>>$OUT/bench.err; python -c "import json;r=json.load(open('$OUT/bench_c1_fp16_lazy.json'));print(r['value'],r['clocks'].get('sclk_mhz',{}).get('mean'),r['clocks'].get('power_w',{}).get('mean'),r['roofline'].get('pipe_counters'),r['roofline'].get('mfma_only'))"
timeout 600 python bench.py --steps 20 --warmup 5 --dtype fp16 --kernel "(FP16, 128, 256, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer+spec_softmax" --no-cpu-baseline --hermetic-reps 0 --no-mfma-roof > $OUT/bench_c1_fp16_spec.json 2>>$OUT/bench.err; python -c "import json;r=json.load(open('$OUT/bench_c1_fp16_spec.json'));print(r['value'],r['clocks'].get('sclk_mhz',{}).get('mean'),r['clocks'].get('power_w',{}).get('mean'),r['roofline'].get('pipe_counters'))"
echo "== c2 / c3"; timeout 900 python bench.py --workload c2 --steps 20 --warmup 5 --no-traffic > $OUT/bench_c2.json 2>>$OUT/bench.err; python -c "import json;r=json.load(open('$OUT/bench_c2.json'));print(r['value'],{k:round(v['tflops']) for k,v in r['per_seq_len'].items()})"
timeout 600 python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --hermetic-reps 0 --no-mfma-roof > $OUT/bench_c3.json 2>>$OUT/bench.err; python -c "import json;r=json.load(open('$OUT/bench_c3.json'));print(r['value'],r['config']['kernel'],r['roofline']['traffic'])"
echo "== done"
