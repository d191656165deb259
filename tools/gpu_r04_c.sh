#!/bin/bash
# round 4, call C: the GPU tier + the bench line's robustness block on the library with the one-outstanding-probe adaptive policy
export PYTHONPATH=$PWD:$PYTHONPATH
OUT=gpurun_out/r04c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.txt 2>&1; tail -25 $OUT/pytest_gpu.txt
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench c1 (no traffic / roof / cpu legs)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-traffic --no-mfma-roof --no-cpu-baseline --hermetic-reps 0 > $OUT/bench_c1.json 2> $OUT/bench.err; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04c/bench_c1.json"))
print("value", d["value"], "sustained", d.get("sustained",{}).get("tflops"), d["speculative"])
for k,v in (d.get("variants") or {}).items():
    if isinstance(v, dict): print("variant", k, {kk: (round(vv,1) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("tflops","tflops_min","tflops_max")}, v.get("ratio_to_default"))
for k,v in (d.get("robustness") or {}).items():
    print("robust", k, {n: round(v[n]["tflops"],1) for n in ("lazy","default","speculative_always") if isinstance(v.get(n),dict) and "tflops" in v[n]}, "redone", v.get("items_redone_by_an_always_speculative_launch"), "/", v.get("items"), v.get("adaptive"), "default/lazy", v.get("default_over_lazy"), v["default"].get("ratio_to_lazy"))
PY
echo "== done"
