TAG=${1:-r06h}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
SPEC="(BF16, 128, 256, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer+spec_softmax"
LAZY="(BF16, 128, 256, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer"
QUICK="--no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof"
echo "== pytest (all)"; timeout -s KILL 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.txt 2>&1; grep -n "^E  \|^FAILED\|passed\|failed" $OUT/pytest_gpu.txt | grep -v "where\|+  " | cut -c1-300 | head -40
echo "== data"
: > $OUT/sink_data.txt
for D in heavy sink; do for T in bf16 fp16; do for K in spec lazy; do
  KK="$SPEC"; [ $K = lazy ] && KK="$LAZY"; [ $T = fp16 ] && KK="${KK/BF16/FP16}"
  timeout -s KILL 600 python bench.py --steps 20 --warmup 5 --data $D --dtype $T --kernel "$KK" $QUICK > $OUT/b.json 2>/dev/null
  python -c "import json;r=json.load(open('$OUT/b.json'));s=r['speculative'];print('%-6s %-5s %-9s %8.1f TFLOP/s   items %d redone %d' % ('$D','$T','$K',r['value'],s['items'],s['items_redone']))" | tee -a $OUT/sink_data.txt
done; done; done
echo "== done"
