#!/bin/bash
# round 4, call B: the GPU tier on the library as committed (rotated plan, adaptive default, jitter twin), the jitter soak,
# the store-policy A/B, and the driver's bench command with its new blocks (variants / robustness / sustained).
export PYTHONPATH=$PWD:$PYTHONPATH
L=flash_attention_from_scratch_amd/lib
OUT=gpurun_out/r04b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.txt 2>&1; tail -25 $OUT/pytest_gpu.txt
echo "== tune64 (store policy)"; timeout 600 $L/tune64 reps=8 > $OUT/tune64_store_policy.txt 2>&1; cut -c1-125 $OUT/tune64_store_policy.txt
echo "== jitter soak 120 s"; FA_HIP_LIB=$PWD/$L/libfa_hip_jitter.so timeout 400 python tools/soak.py 120 13 > $OUT/soak_jitter.txt 2>&1; tail -3 $OUT/soak_jitter.txt
echo "== bench c1 (driver command)"; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c1.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench_c1.json; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04b/bench_c1.json"))
print("value", d["value"], "frac", d["roofline"]["frac"], "wcpm", d.get("wave_cycles_per_mfma"), "busy", d.get("mfma_busy_frac_of_wave_time"), "fpc", d.get("frac_of_peak_at_measured_clock"))
print("sustained", d.get("sustained"))
for k,v in (d.get("variants") or {}).items():
    if isinstance(v, dict): print("variant", k, {kk: (round(vv,1) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("tflops","tflops_min","tflops_max")}, v.get("ratio_to_default"))
for k,v in (d.get("robustness") or {}).items():
    print("robust", k, {n: round(v[n]["tflops"],1) for n in ("lazy","default","speculative_always") if isinstance(v.get(n),dict) and "tflops" in v[n]}, "redone", v.get("items_redone_by_an_always_speculative_launch"), "/", v.get("items"), v.get("adaptive"), "default/lazy", v.get("default_over_lazy"))
PY
echo "== done"
