#!/bin/bash
# round 4, closing call: GPU tier + smoke on the library as committed, then the soaks (every device variant, the many-items
# walk, the same under the jitter build)
export PYTHONPATH=$PWD:$PYTHONPATH
L=flash_attention_from_scratch_amd/lib
OUT=gpurun_out/r04f
mkdir -p $OUT
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
echo "== pytest"; timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
echo "== soak, every variant, 150 s"; timeout 400 python tools/soak.py 150 13 all > $OUT/soak_all.txt 2>&1; tail -2 $OUT/soak_all.txt
echo "== soak, persistent kernel, 60 s under the jitter build (another seed)"; FA_HIP_LIB=$PWD/$L/libfa_hip_jitter.so timeout 300 python tools/soak.py 60 29 > $OUT/soak_jitter2.txt 2>&1; tail -2 $OUT/soak_jitter2.txt
echo "== many-items walk, 40 s product + 40 s jitter"; timeout 200 python tools/soak_many_items.py 40 5 > $OUT/soak_many.txt 2>&1; tail -1 $OUT/soak_many.txt; FA_HIP_LIB=$PWD/$L/libfa_hip_jitter.so timeout 200 python tools/soak_many_items.py 40 6 >> $OUT/soak_many.txt 2>&1; tail -1 $OUT/soak_many.txt
echo "== c client"; tail -1 $OUT/pytest_gpu.txt
echo "== done"
