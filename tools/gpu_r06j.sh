# Round 6: the alternating walk against the shipped kernel in one process (tune64, variant -7, S = 16384), its GPU test, the
# item traces with the chip-wide counter.   bash tools/gpu_r06j.sh [tag]
TAG=${1:-r06j}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
L=flash_attention_from_scratch_amd/lib
echo "== pytest (the two tests touched)"; timeout -s KILL 1200 python -m pytest tests -m gpu -q --timeout 900 -k "alternate or beyond_ordinal" > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
echo "== tune64 x2"; timeout 1200 $L/tune64 reps=8 > $OUT/tune64.txt 2>&1; timeout 1200 $L/tune64 reps=8 > $OUT/tune64_again.txt 2>&1; grep -h "S=16384\|S= 4096" $OUT/tune64.txt | cut -c1-200; echo "-- again"; grep -h "S=16384" $OUT/tune64_again.txt | cut -c1-200
echo "== trace64_items"; for a in "512 16 16" "1024 16 16" "4096 4 16"; do timeout 120 $L/trace64_items $a 2>&1 | grep -E "^==|mean over|^realtime" | tail -3; done > $OUT/trace64_items.txt; cut -c1-420 $OUT/trace64_items.txt
echo "== done"
