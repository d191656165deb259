# Round 6, closing lease: the whole GPU tier and long soaks on the library as committed (product and jitter build; the
# long-sequence form is in the soak's mix).   bash tools/gpu_r06m.sh [tag]
TAG=${1:-r06m}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
L=flash_attention_from_scratch_amd/lib
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
echo "== pytest"; timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
echo "== soak 300 s"; timeout 600 python tools/soak.py 300 101 > $OUT/soak_300s.txt 2>&1; tail -1 $OUT/soak_300s.txt
echo "== soak 300 s, jitter build"; FA_HIP_LIB=$PWD/$L/libfa_hip_jitter.so timeout 600 python tools/soak.py 300 103 > $OUT/soak_jitter_300s.txt 2>&1; tail -1 $OUT/soak_jitter_300s.txt
echo "== soak 120 s, every variant"; timeout 400 python tools/soak.py 120 105 all > $OUT/soak_all_120s.txt 2>&1; tail -1 $OUT/soak_all_120s.txt
echo "== bench (driver's command)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c1.json 2> $OUT/bench.err; cut -c1-260 $OUT/bench_c1.json
echo "== done"
