# Round 6: the eight-wave ring form as the product's ring form of (128, 64, 4)+buffer: GPU tier, check tools, ring soak (product + jitter),
# the shape at C1 under the driver's protocol, tune-free timing.   bash tools/gpu_r06p.sh [tag]
TAG=${1:-r06p}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
L=flash_attention_from_scratch_amd/lib
echo "== pytest"; timeout -s KILL 2400 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.txt 2>&1; grep -n "^E  \|^FAILED\|passed\|failed" $OUT/pytest_gpu.txt | grep -v "where\|+  " | cut -c1-260 | head -40
echo "== check_nw8"; timeout -s KILL 300 $L/check_nw8 > $OUT/check_nw8.txt 2>&1; grep -v "^  S=" $OUT/check_nw8.txt | tail -18 | cut -c1-200
echo "== soak ring 60 s + 60 s jitter"; timeout 300 python tools/soak.py 60 9 ring > $OUT/soak_ring.txt 2>&1; FA_HIP_LIB=$PWD/$L/libfa_hip_jitter.so timeout 300 python tools/soak.py 60 11 ring >> $OUT/soak_ring.txt 2>&1; grep "^soak\|FAIL" $OUT/soak_ring.txt | tail -4
echo "== ring_c1"
RB="(BF16, 128, 128, 64, 4): async+eager+swizzled+load_0_0_0_tiles"
for rep in 1 2; do for kk in "$RB+buffer" "$RB+buffer+spec_softmax" "$RB" "${RB/BF16/FP16}+buffer" "${RB/BF16/FP16}+buffer+spec_softmax"; do
  dt=bf16; case "$kk" in "(FP16"*) dt=fp16;; esac
  python bench.py --steps 20 --warmup 5 --dtype $dt --kernel "$kk" --no-variants --no-traffic --no-cpu-baseline --no-mfma-roof --hermetic-reps 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$rep', d['config']['kernel'].ljust(88), '%.1f TFLOP/s  sustained %.1f  hermetic %.1f' % (d['value'], d.get('sustained', {}).get('tflops', 0), d.get('protocols', {}).get('hermetic', {}).get('tflops', 0)))"
done; done > $OUT/ring_c1.txt; cat $OUT/ring_c1.txt
echo "== bench c1 quick (default unchanged?)"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof --no-variants > $OUT/bench_c1.json 2>/dev/null; cut -c1-200 $OUT/bench_c1.json
echo "== done"
