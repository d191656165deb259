TAG=${1:-r06a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
L=flash_attention_from_scratch_amd/lib
SPEC="(BF16, 128, 256, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer+spec_softmax"
LAZY="(BF16, 128, 256, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer"
QUICK="--no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof"
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
echo "== queue tests first"; timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x -k "redo_queue or second_pass or stateless or limit_and_large or ring_form or seam_golden" > $OUT/pytest_queue.txt 2>&1; tail -25 $OUT/pytest_queue.txt
echo "== tune64"; timeout 500 $L/tune64 reps=6 > $OUT/tune64.txt 2>&1; grep -h "S= 4096\|S=  512" $OUT/tune64.txt | cut -c1-150
echo "== check_qt1"; timeout 120 $L/check_qt1 > $OUT/check_qt1.txt 2>&1; tail -3 $OUT/check_qt1.txt
echo "== data"
: > $OUT/sink_data.txt
for D in sink heavy; do for T in bf16 fp16; do for K in spec lazy; do
  KK="$SPEC"; [ $K = lazy ] && KK="$LAZY"; [ $T = fp16 ] && KK="${KK/BF16/FP16}"
  timeout 600 python bench.py --steps 20 --warmup 5 --data $D --dtype $T --kernel "$KK" $QUICK > $OUT/b.json 2>/dev/null
  python -c "import json;r=json.load(open('$OUT/b.json'));s=r['speculative'];print('%-6s %-5s %-9s %8.1f TFLOP/s   items %d redone %d' % ('$D','$T','$K',r['value'],s['items'],s['items_redone']))" | tee -a $OUT/sink_data.txt
done; done; done
echo "== pytest (all)"; timeout 2700 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.txt 2>&1; tail -30 $OUT/pytest_gpu.txt
echo "== bench c1"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c1.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench_c1.json; tail -2 $OUT/bench.err
echo "== done"
