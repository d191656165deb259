# Round 6, one lease: the long-sequence form that alternates the K / V direction (C3 traffic, its GPU test), the S = 512 launch
# breakdown with the chip-wide 100-MHz counter, the two S = 512 knobs (tune64: 256 / 512 / 768), heavy-tailed data after the
# guard's cheaper rescue.   bash tools/gpu_r06i.sh [tag] [skip-tests]
TAG=${1:-r06i}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
L=flash_attention_from_scratch_amd/lib
SPEC="(BF16, 128, 256, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer+spec_softmax"
LAZY="(BF16, 128, 256, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer"
QUICK="--no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof"
if [ -z "$2" ]; then
echo "== pytest (all)"; timeout -s KILL 2400 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.txt 2>&1; grep -n "^E  \|^FAILED\|passed\|failed" $OUT/pytest_gpu.txt | grep -v "where\|+  " | cut -c1-300 | head -40
fi
echo "== c3 traffic"; timeout 900 python tools/c3_traffic.py > $OUT/c3_traffic.txt 2>&1; grep "read bytes\|FETCH\|==" $OUT/c3_traffic.txt | cut -c1-200
echo "== bench c3 (default, with the in-run traffic pass) and lazy"
timeout 900 python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --hermetic-reps 0 --no-mfma-roof --no-variants > $OUT/bench_c3.json 2>/dev/null
python -c "import json;r=json.load(open('$OUT/bench_c3.json'));rf=r['roofline'];print('c3 %.1f TFLOP/s  traffic %s / algorithmic %s = %.3f  kv_walk_alternates %s' % (r['value'], rf['traffic'], rf['algorithmic_bytes'], (rf['traffic'] or 0)/rf['algorithmic_bytes'], r['config'].get('kv_walk_alternates')))"
timeout 900 python bench.py --workload c3 --steps 10 --warmup 3 $QUICK --kernel "${LAZY/BF16/FP16}" > $OUT/bench_c3_lazy.json 2>/dev/null; cut -c1-160 $OUT/bench_c3_lazy.json
echo "== data: heavy / sink, spec vs lazy"
: > $OUT/sink_data.txt
for D in heavy sink; do for T in bf16 fp16; do for K in spec lazy spec lazy; do
  KK="$SPEC"; [ $K = lazy ] && KK="$LAZY"; [ $T = fp16 ] && KK="${KK/BF16/FP16}"
  timeout -s KILL 600 python bench.py --steps 20 --warmup 5 --data $D --dtype $T --kernel "$KK" $QUICK > $OUT/b.json 2>/dev/null
  python -c "import json;r=json.load(open('$OUT/b.json'));s=r['speculative'];print('%-6s %-5s %-9s %8.1f TFLOP/s   items %d redone %d' % ('$D','$T','$K',r['value'],s['items'],s['items_redone']))" | tee -a $OUT/sink_data.txt
done; done; done
echo "== s512 floor"; timeout 900 python tools/s512_floor.py > $OUT/s512_launch_breakdown.txt 2>&1; cut -c1-420 $OUT/s512_launch_breakdown.txt; timeout 900 python tools/s512_floor.py --seq 1024 > $OUT/s1024_launch_breakdown.txt 2>&1; grep "dispatch records\|event-timed\|realtime" $OUT/s1024_launch_breakdown.txt | cut -c1-420
echo "== trace64_tl S=512"; timeout 120 $L/trace64_tl 512 16 2>&1 | grep -E "^==|^mean|^spread|^wg   0" > $OUT/trace64_timeline_s512.txt; cut -c1-260 $OUT/trace64_timeline_s512.txt
echo "== tune64 x2"; timeout 900 $L/tune64 reps=8 > $OUT/tune64.txt 2>&1; timeout 900 $L/tune64 reps=8 > $OUT/tune64_again.txt 2>&1; grep -h "S=  512\|S= 1024\|S= 4096" $OUT/tune64.txt | cut -c1-150; echo "-- again"; grep -h "S=  512\|S= 1024" $OUT/tune64_again.txt | cut -c1-150
echo "== bench c1 quick x2"; for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 $QUICK --no-variants > $OUT/bench_c1_$i.json 2>/dev/null; cut -c1-200 $OUT/bench_c1_$i.json; done
echo "== done"
