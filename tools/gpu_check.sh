#!/bin/bash
# After a kernel change: GPU suite, soaks, the interleaved tune64 table, the C2 sweep.  Usage: bash tools/gpu_check.sh [tag]
OUT=gpurun_out/${1:-check}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
L=flash_attention_from_scratch_amd/lib
echo "== pytest"; timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
echo "== soaks"; timeout 400 python tools/soak.py 90 7 all > $OUT/soak.txt 2>&1; tail -1 $OUT/soak.txt
timeout 300 python tools/soak_many_items.py 45 3 > $OUT/soak_many_items.txt 2>&1; tail -1 $OUT/soak_many_items.txt
echo "== tune64"; timeout 600 $L/tune64 reps=10 > $OUT/tune64.txt 2>&1; grep -E "default|as before|TIMING|lazy" $OUT/tune64.txt | cut -c1-40,100-220
echo "== c2"; timeout 900 python bench.py --workload c2 --steps 20 --warmup 5 --no-traffic > $OUT/bench_c2.json 2>/dev/null; python -c "import json;r=json.load(open('$OUT/bench_c2.json'));print(r['value'],{k:round(v['tflops']) for k,v in r['per_seq_len'].items()})"
echo "== c1"; timeout 600 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof > $OUT/b.json 2>/dev/null; python -c "import json;r=json.load(open('$OUT/b.json'));print(r['value'])"
