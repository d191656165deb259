#!/bin/bash
# One gpurun call worth of evidence: probe -> smoke -> pytest -m gpu -> bench lines -> sweeps -> rocprof -> PMC.
# Usage (from the repo root, on the GPU box):  bash tools/gpu_round.sh [tag]
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
echo "== probe"; timeout 120 ./flash_attention_from_scratch_amd/lib/layout_probe > $OUT/probe.txt 2>&1; tail -3 $OUT/probe.txt
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -3 $OUT/smoke.txt
echo "== pytest"; timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
echo "== bench c1 (the driver's command: clocks, per-launch distribution, in-run PMC traffic, cpu baseline)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c1.json 2> $OUT/bench.err; cut -c1-700 $OUT/bench_c1.json; tail -2 $OUT/bench.err
echo "== bench c1, 2000 steps (sustained)"; timeout 600 python bench.py --steps 2000 --warmup 10 --no-cpu-baseline --no-traffic > $OUT/bench_c1_sustained.json 2>/dev/null; cut -c1-200 $OUT/bench_c1_sustained.json
echo "== bench c1, no preconditioning (cold clocks)"; timeout 600 python bench.py --steps 20 --warmup 5 --precondition-ms 0 --no-cpu-baseline --no-traffic > $OUT/bench_c1_cold.json 2>/dev/null; cut -c1-200 $OUT/bench_c1_cold.json
echo "== bench c1 --hermetic"; timeout 600 python bench.py --steps 50 --warmup 10 --hermetic --no-cpu-baseline --no-traffic > $OUT/bench_c1_hermetic.json 2>/dev/null; cut -c1-200 $OUT/bench_c1_hermetic.json
echo "== bench c1, lazy-rescale build (optimized_softmax = False)"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --kernel "(BF16, 128, 256, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer" > $OUT/bench_c1_lazy.json 2>/dev/null; cut -c1-200 $OUT/bench_c1_lazy.json
for W in c3 c4; do echo "== bench $W"; timeout 900 python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$W.json 2>/dev/null; cut -c1-200 $OUT/bench_$W.json; done
echo "== bench c2 (seq sweep, harmonic mean)"; timeout 900 python bench.py --workload c2 --steps 20 --warmup 5 > $OUT/bench_c2.json 2>/dev/null; cut -c1-200 $OUT/bench_c2.json
echo "== bench --gpus 2, self-launched (gloo; both ranks on this box's one GPU: plumbing only)"; timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/bench_n2_selflaunch.json 2> $OUT/bench_n2.err; cut -c1-200 $OUT/bench_n2_selflaunch.json
echo "== wideners"; timeout 600 python flash_attention_from_scratch_amd/tools/bench_wideners.py > $OUT/wideners.txt 2>/dev/null; cat $OUT/wideners.txt
echo "== tune64 (plan variants / ablations of the persistent kernel)"; timeout 600 ./flash_attention_from_scratch_amd/lib/tune64 reps=6 > $OUT/tune64.txt 2>&1; grep "S= 4096" $OUT/tune64.txt
echo "== mfma_energy"; timeout 300 ./flash_attention_from_scratch_amd/lib/mfma_energy > $OUT/mfma_energy.txt 2>&1; cat $OUT/mfma_energy.txt
echo "== seqsweep"; bash tools/gpu_seqsweep.sh $TAG/seq > /dev/null 2>&1
echo "== sweeps"; for K in native tune; do KERNELS=$K timeout 900 python flash_attention_from_scratch_amd/tools/pt_bench.py --seq_lens 4096 --batch 4 --num_repeats 20 --num_warmups 5 > $OUT/sweep_${K}_c1.csv 2> $OUT/sweep.err; done; head -4 $OUT/sweep_native_c1.csv | cut -c1-160; tail -2 $OUT/sweep.err
echo "== rocprof --kernel-trace --stats of the bench command"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o fa -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $OUT/rocprof_bench.json 2> $OUT/rocprof.err; tail -2 $OUT/rocprof.err; for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -4 $f; cp $f $OUT/rocprof_kernel_stats.csv; done
echo "== pmc"; BEST=$(python -c "from flash_helpers.kernel_configs import best_config; print(best_config().short_form())"); bash tools/gpu_pmc.sh $TAG/pmc "$BEST" "(BF16, 128, 256, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer" > $OUT/pmc.log 2>&1; tail -60 $OUT/pmc/pmc_summary.txt; cat $OUT/pmc/pmc_traffic.json
echo "== done"
