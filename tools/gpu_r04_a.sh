#!/bin/bash
# round 4, call A: the rotated plan (flash_attention_from_scratch_amd/csrc/fa_fwd_kernel64.hpp, make_plan64 rot_k).
# lib/libfa_old.so = the library as round 3 left it, lib/libfa_hip.so = the same source built with -DFA_ROT_DEFAULT=5.
export PYTHONPATH=$PWD:$PYTHONPATH
L=flash_attention_from_scratch_amd/lib
OUT=gpurun_out/r04a
mkdir -p $OUT
echo "== tune64"; timeout 600 $L/tune64 reps=8 > $OUT/tune64.txt 2>&1; grep -E "S= 4096|S=  512" $OUT/tune64.txt
echo "== trace64 groups (visit 20)"; for r in 0 5; do timeout 120 $L/trace64_rot$r 2>&1 | grep -E "visit 20 wave 1|visit 62 wave 1|visit  0 wave 1" | sed "s/^/rot$r /"; done | tee $OUT/trace64_groups.txt
echo "== trace64_items"; for a in "512 16 16" "1024 16 16" "4096 4 16"; do for b in trace64_items trace64_items_rot5; do timeout 120 $L/$b $a 2>&1 | grep -E "^==|mean over" | tail -2 | sed "s/^/$b /"; done; done | cut -c1-400 | tee $OUT/trace64_items.txt
echo "== pytest (rotated library)"; timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.txt 2>&1; tail -15 $OUT/pytest_gpu.txt
echo "== bench A/B"
cp $L/libfa_hip.so $L/libfa_new.so
for rep in 1 2 3; do for w in old new; do
  cp $L/libfa_$w.so $L/libfa_hip.so
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', f\"{d['value']:8.1f} TF  {d['ms_per_step']:.4f} ms\", d.get('clocks',{}).get('sclk_mhz',{}).get('mean'), d.get('clocks',{}).get('power_w',{}).get('mean'))"
done; done | tee $OUT/ab.txt
cp $L/libfa_new.so $L/libfa_hip.so
echo "== c2 new"; timeout 600 python bench.py --workload c2 --steps 20 --warmup 5 --no-traffic --no-cpu-baseline 2>/dev/null > $OUT/bench_c2_new.json; cut -c1-300 $OUT/bench_c2_new.json
cp $L/libfa_old.so $L/libfa_hip.so
echo "== c2 old"; timeout 600 python bench.py --workload c2 --steps 20 --warmup 5 --no-traffic --no-cpu-baseline 2>/dev/null > $OUT/bench_c2_old.json; cut -c1-300 $OUT/bench_c2_old.json
cp $L/libfa_new.so $L/libfa_hip.so
echo "== done"
