# Round 6: the alternating K / V walk against the plain walk on C3 ITSELF (fp16 B = 2 H = 32 S = 16384) and on C2's S = 16384 shape,
# same lease, alternating processes (FA_HIP_NO_ALT: the launcher's measurement switch).   bash tools/gpu_r06k.sh [tag]
TAG=${1:-r06k}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
Q="--no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof --no-variants"
: > $OUT/c3_alt_ab.txt
for rep in 1 2 3 4; do for mode in alt plain; do
  if [ $mode = plain ]; then export FA_HIP_NO_ALT=1; else unset FA_HIP_NO_ALT; fi
  timeout 600 python bench.py --workload c3 --steps 30 --warmup 5 $Q > $OUT/b.json 2>/dev/null
  python -c "import json;r=json.load(open('$OUT/b.json'));print('c3 fp16 B=2 H=32 S=16384  %-5s  %.1f TFLOP/s  %.4f ms  sclk %s' % ('$mode', r['value'], r['ms_per_step'], r.get('clocks',{}).get('sclk_mhz',{}).get('mean')))" | tee -a $OUT/c3_alt_ab.txt
done; done
unset FA_HIP_NO_ALT
echo "== done"
