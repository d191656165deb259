#!/bin/bash
# One gpurun call worth of evidence: probe -> smoke -> pytest -m gpu -> bench -> sweep -> rocprof.
# Usage (from the repo root, on the GPU box):  bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export PYTHONPATH=$PWD:$PYTHONPATH
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
echo "== probe"; timeout 120 ./flash_attention_from_scratch_amd/lib/layout_probe > $OUT/probe.txt 2>&1; tail -8 $OUT/probe.txt
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -5 $OUT/smoke.txt
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.txt 2>&1; tail -15 $OUT/pytest_gpu.txt
echo "== bench"; timeout 600 python bench.py --steps 30 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -3 $OUT/bench.err
echo "== bench c3"; timeout 600 python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c3.json 2>/dev/null; cut -c1-300 $OUT/bench_c3.json
echo "== bench c4 (one GPU's shard)"; timeout 600 python bench.py --workload c4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c4.json 2>/dev/null; cut -c1-300 $OUT/bench_c4.json
echo "== bench c2 (seq sweep, harmonic mean)"; timeout 600 python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c2.json 2>/dev/null; cut -c1-400 $OUT/bench_c2.json
echo "== bench, N=2 code path on one GPU (gloo, both ranks on cuda:0; the timings mean nothing)"; timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --dist-backend gloo > $OUT/bench_n2_gloo.json 2> $OUT/bench_n2_gloo.err; cut -c1-260 $OUT/bench_n2_gloo.json; tail -2 $OUT/bench_n2_gloo.err
echo "== wideners"; timeout 600 python flash_attention_from_scratch_amd/tools/bench_wideners.py > $OUT/wideners.txt 2>/dev/null; cat $OUT/wideners.txt
echo "== dvfs"; timeout 300 python flash_attention_from_scratch_amd/tools/dvfs_probe.py > $OUT/dvfs_probe.txt 2>/dev/null; cat $OUT/dvfs_probe.txt
echo "== seqsweep"; bash tools/gpu_seqsweep.sh $TAG/seq > /dev/null 2>&1
echo "== sweep"; KERNELS=native timeout 900 python flash_attention_from_scratch_amd/tools/pt_bench.py --seq_lens 4096 --batch 4 --num_repeats 20 --num_warmups 5 > $OUT/sweep_native_c1.csv 2> $OUT/sweep.err; head -6 $OUT/sweep_native_c1.csv | cut -c1-160; tail -3 $OUT/sweep.err
echo "== rocprof"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o fa -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/rocprof_bench.json 2> $OUT/rocprof.err; tail -2 $OUT/rocprof.err; find $OUT/prof -name "*stats*" | head; for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -5 $f; done
echo "== pmc"; BEST=$(python -c "from flash_helpers.kernel_configs import best_config; print(best_config().short_form())"); bash tools/gpu_pmc.sh $TAG/pmc "$BEST" > $OUT/pmc.log 2>&1; tail -40 $OUT/pmc/pmc_summary.txt; cat $OUT/pmc/pmc_traffic.json
echo "== done"
