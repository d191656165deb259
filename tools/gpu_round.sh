#!/bin/bash
# THE script for the GPU box: one gpurun call, one lease, everything profiles/<round>/ holds.
#   bash tools/gpu_round.sh [tag] [section ...]        (from the repo root; no section = all of them, in this order)
# sections:  tests      probe, smoke(), pytest -m gpu
#            bench      the driver's command (full line) + rocprofv3 --kernel-trace --stats of the same command (same lease),
#                       2000 steps (sustained), no preconditioning (cold), lazy build, pre-scaled Q, fp16 + fp16 lazy
#            data       non-Gaussian data (sink / heavy) x {always speculative, adaptive, lazy} x {bf16, fp16}
#            workloads  c3, c4 (one GPU's shard), c3 lazy, the c2 sweep, --gpus 2 and the eight-launchers-one-host rehearsal, self-launched;
#                       where an S = 512 launch goes (tools/s512_floor.py)
#            wideners   causal / ragged timings, the masked build with nothing masked (tools/masked_probe.py)
#            tune       tune64: this tree's variants and the PREVIOUS round's kernel in one process; item / seam traces
#            sweeps     pt_bench over the native configs and the reference's 80 (KERNELS=tune)
#            pmc        counter passes: default / lazy / pre-scaled Q (bf16) and default / lazy (fp16) at C1; C3 traffic by size
#            soak       60 s of random launches of every variant, the same under the jitter build, jitter_check
# Everything runs on ONE lease: bench_c1.json and rocprof_kernel_stats.csv are the same box.  (The one-off scripts of rounds
# 1-4 are in the git history: `git log --diff-filter=D --name-only -- tools/`.)
TAG=${1:-r06}; shift
SECTIONS="${*:-tests bench data workloads wideners tune sweeps pmc soak}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
L=flash_attention_from_scratch_amd/lib
SPEC="(BF16, 128, 256, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer+spec_softmax"
ADAPT="$SPEC+adaptive"   # opt-in since round 6 (best_config() = $SPEC: stateless)
LAZY="(BF16, 128, 256, 64, 4): async+eager+swizzled+load_0_0_0_tiles+buffer"
QUICK="--no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof"
has() { [[ " $SECTIONS " == *" $1 "* ]]; }

if has tests; then
echo "== probe"; timeout 120 $L/layout_probe > $OUT/probe.txt 2>&1; tail -3 $OUT/probe.txt
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -3 $OUT/smoke.txt
echo "== pytest"; timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt > $OUT/pytest_gpu_tail.txt
fi
if has bench; then
echo "== bench c1 (the driver's command: value, hermetic protocol, MFMA-only roof, clocks, in-run PMC traffic + pipe counters, variants, robustness, cpu baseline)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c1.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench_c1.json; tail -2 $OUT/bench.err
echo "== rocprof --kernel-trace --stats of the bench command (same lease as bench_c1.json)"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o fa -- python bench.py --steps 20 --warmup 5 $QUICK --no-variants > $OUT/rocprof_bench.json 2> $OUT/rocprof.err; tail -2 $OUT/rocprof.err; for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -4 $f; cp $f $OUT/rocprof_kernel_stats.csv; done
echo "== bench c1, 2000 steps (sustained)"; timeout 600 python bench.py --steps 2000 --warmup 10 $QUICK --no-variants > $OUT/bench_c1_sustained.json 2>/dev/null; cut -c1-200 $OUT/bench_c1_sustained.json
echo "== bench c1, no preconditioning (cold clocks)"; timeout 600 python bench.py --steps 20 --warmup 5 --precondition-ms 0 $QUICK --no-variants > $OUT/bench_c1_cold.json 2>/dev/null; cut -c1-200 $OUT/bench_c1_cold.json
echo "== bench c1, lazy-rescale build"; timeout 600 python bench.py --steps 20 --warmup 5 $QUICK --kernel "$LAZY" > $OUT/bench_c1_lazy.json 2>/dev/null; cut -c1-200 $OUT/bench_c1_lazy.json
echo "== bench c1, pre-scaled Q (opt-in)"; timeout 600 python bench.py --steps 20 --warmup 5 $QUICK --kernel "$SPEC+prescaled_q" > $OUT/bench_c1_prescaled_q.json 2>/dev/null; cut -c1-200 $OUT/bench_c1_prescaled_q.json
echo "== bench c1 fp16 (default) with pipe counters, and fp16 lazy"
timeout 600 python bench.py --steps 20 --warmup 5 --dtype fp16 --no-cpu-baseline --hermetic-reps 0 --no-variants > $OUT/bench_c1_fp16.json 2>/dev/null; cut -c1-200 $OUT/bench_c1_fp16.json
timeout 600 python bench.py --steps 20 --warmup 5 --dtype fp16 $QUICK --kernel "${LAZY/BF16/FP16}" > $OUT/bench_c1_fp16_lazy.json 2>/dev/null; cut -c1-200 $OUT/bench_c1_fp16_lazy.json
fi
if has data; then
echo "== non-Gaussian data: what the speculative softmax's second pass costs (sink: +12 nats at the first 4 keys; heavy: Student-t K)"
: > $OUT/sink_data.txt
for D in randn sink heavy; do for T in bf16 fp16; do for K in spec adaptive lazy; do
  KK="$SPEC"; [ $K = lazy ] && KK="$LAZY"; [ $K = adaptive ] && KK="$ADAPT"; [ $T = fp16 ] && KK="${KK/BF16/FP16}"
  timeout 600 python bench.py --steps 20 --warmup 5 --data $D --dtype $T --kernel "$KK" $QUICK > $OUT/b.json 2>/dev/null
  python -c "import json;r=json.load(open('$OUT/b.json'));s=r['speculative'];print('%-6s %-5s %-9s %8.1f TFLOP/s   items %d redone %d (%.1f %%) in one more launch; adaptive: %s' % ('$D','$T','$K',r['value'],s['items'],s['items_redone'],100*s['second_pass_fraction'],s.get('adaptive')))" | tee -a $OUT/sink_data.txt
done; done; done
fi
if has workloads; then
for W in c3 c4; do echo "== bench $W"; timeout 900 python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --hermetic-reps 0 --no-mfma-roof --no-variants > $OUT/bench_$W.json 2>/dev/null; cut -c1-200 $OUT/bench_$W.json; done
echo "== bench c3 with the lazy rescale"; timeout 900 python bench.py --workload c3 --steps 10 --warmup 3 $QUICK --kernel "${LAZY/BF16/FP16}" > $OUT/bench_c3_lazy.json 2>/dev/null; cut -c1-200 $OUT/bench_c3_lazy.json
echo "== bench c2 (seq sweep, harmonic mean; per-seq_len rooflines)"; timeout 900 python bench.py --workload c2 --steps 20 --warmup 5 > $OUT/bench_c2.json 2>/dev/null; cut -c1-200 $OUT/bench_c2.json
echo "== bench --gpus 2, self-launched (gloo; both ranks on this box's one GPU: plumbing only -- NUMA pinning, affinities in the line)"; timeout 600 python bench.py --gpus 2 --warmup 2 > $OUT/bench_n2_selflaunch.json 2> $OUT/bench_n2.err; cut -c1-200 $OUT/bench_n2_selflaunch.json
echo "== bench --gpus 8 --workload c4 --batch-per-rank 1, self-launched: eight launchers on ONE host against this box's one GPU (host us per launch, affinity, barrier latency per rank)"; timeout 900 python bench.py --gpus 8 --workload c4 --batch-per-rank 1 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --hermetic-reps 0 --no-mfma-roof --no-variants > $OUT/bench_n8_selflaunch.json 2> $OUT/bench_n8.err; cut -c1-300 $OUT/bench_n8_selflaunch.json; tail -2 $OUT/bench_n8.err
echo "== where an S = 512 launch goes: kernel duration and gap to the next dispatch (rocprofv3 --kernel-trace), event-timed interval, workgroup walk"; timeout 900 python tools/s512_floor.py > $OUT/s512_launch_breakdown.txt 2>&1; cat $OUT/s512_launch_breakdown.txt | cut -c1-300; timeout 900 python tools/s512_floor.py --seq 1024 > $OUT/s1024_launch_breakdown.txt 2>&1; grep "dispatch records\|event-timed" $OUT/s1024_launch_breakdown.txt | cut -c1-300
fi
if has wideners; then
echo "== wideners"; timeout 600 python flash_attention_from_scratch_amd/tools/bench_wideners.py > $OUT/wideners.txt 2>/dev/null; cat $OUT/wideners.txt
echo "== masked build with nothing masked"; timeout 300 python tools/masked_probe.py 4096 4 16 > $OUT/masked_probe.txt 2>/dev/null; cut -c1-110 $OUT/masked_probe.txt
fi
if has tune; then
echo "== tune64 (this tree's variants + the previous round's kernel, one process, interleaved; twice)"; timeout 600 $L/tune64 reps=8 > $OUT/tune64.txt 2>&1; timeout 600 $L/tune64 reps=8 > $OUT/tune64_again.txt 2>&1; grep -h "S= 4096\|S=  512" $OUT/tune64.txt | cut -c1-140
echo "== check_qt1: the one-Q-tile-per-wave forms (lazy, speculative) against the 64-row lazy kernel, bit for bit"; timeout 120 $L/check_qt1 > $OUT/check_qt1.txt 2>&1; tail -3 $OUT/check_qt1.txt
echo "== check_nw8: the eight-wave ring form (the product's ring form of (128, 64, 4)+buffer since round 6) against the 64-row lazy kernel, and timed"; timeout 300 $L/check_nw8 > $OUT/check_nw8.txt 2>&1; grep -v "^  S=" $OUT/check_nw8.txt | tail -17 | cut -c1-160
echo "== trace64_items / seam"; for a in "512 16 16" "1024 16 16" "4096 4 16"; do timeout 120 $L/trace64_items $a 2>&1 | grep -E "^==|mean over|^realtime" | tail -3; done > $OUT/trace64_items.txt; for a in "512 16 16" "4096 4 16"; do timeout 120 $L/trace64_seam $a 2>&1 | grep -E "first seam" | tail -1 >> $OUT/trace64_items.txt; done; cut -c1-300 $OUT/trace64_items.txt
echo "== trace64_tl: the prologue and the visits of a two-item walk (S = 512), warm and flushed"; timeout 120 $L/trace64_tl 512 16 2>&1 | grep -E "^==|^mean|^spread|^wg   0" > $OUT/trace64_timeline_s512.txt; cut -c1-260 $OUT/trace64_timeline_s512.txt
echo "== mfma_energy"; timeout 300 $L/mfma_energy > $OUT/mfma_energy.txt 2>&1; $L/mfma_energy quick >> $OUT/mfma_energy.txt 2>&1; tail -6 $OUT/mfma_energy.txt
fi
if has sweeps; then
echo "== the reference's winning shape at C1 under the driver's protocol: ring form lazy / speculative, and the shape without +buffer (compiler-scheduled body)"
RB="(BF16, 128, 128, 64, 4): async+eager+swizzled+load_0_0_0_tiles"
for rep in 1 2; do for kk in "$RB+buffer" "$RB+buffer+spec_softmax" "$RB" "${RB/BF16/FP16}+buffer" "${RB/BF16/FP16}+buffer+spec_softmax"; do
  dt=bf16; case "$kk" in "(FP16"*) dt=fp16;; esac
  python bench.py --steps 20 --warmup 5 --dtype $dt --kernel "$kk" --no-variants --no-traffic --no-cpu-baseline --no-mfma-roof --hermetic-reps 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$rep', d['config']['kernel'].ljust(88), '%.1f TFLOP/s  sustained %.1f  hermetic %.1f' % (d['value'], d.get('sustained', {}).get('tflops', 0), d.get('protocols', {}).get('hermetic', {}).get('tflops', 0)))"
done; done > $OUT/ring_c1.txt; cat $OUT/ring_c1.txt
echo "== sweeps (native; the reference's 80 configs with the reference's meaning of opt_softmax)"
KERNELS=native timeout 900 python flash_attention_from_scratch_amd/tools/pt_bench.py --seq_lens 4096 --batch 4 --num_repeats 20 --num_warmups 5 > $OUT/sweep_native_c1.csv 2> $OUT/sweep.err
KERNELS=tune timeout 900 python flash_attention_from_scratch_amd/tools/pt_bench.py --seq_lens 4096 --batch 4 --num_repeats 20 --num_warmups 5 > $OUT/sweep_tune_c1.csv 2>> $OUT/sweep.err
head -4 $OUT/sweep_native_c1.csv | cut -c1-160; tail -2 $OUT/sweep.err
fi
if has pmc; then
echo "== pmc, bf16: default / lazy / pre-scaled Q"; bash tools/gpu_pmc.sh $TAG/pmc "$SPEC" "$LAZY" "$SPEC+prescaled_q" > $OUT/pmc.log 2>&1; tail -30 $OUT/pmc/pmc_summary.txt; cat $OUT/pmc/pmc_traffic.json
echo "== pmc, fp16: default / lazy (VERDICT r04 7: where does fp16 trail bf16 -- cycles or clock?)"; bash tools/gpu_pmc.sh $TAG/pmc_fp16 "${SPEC/BF16/FP16}" "${LAZY/BF16/FP16}" > $OUT/pmc_fp16.log 2>&1; tail -30 $OUT/pmc_fp16/pmc_summary.txt
python tools/fp16_vs_bf16.py $OUT > $OUT/fp16_vs_bf16.txt 2>&1; cat $OUT/fp16_vs_bf16.txt
echo "== c3 traffic by request size"; timeout 600 python tools/c3_traffic.py > $OUT/c3_traffic.txt 2>&1; grep "read bytes\|FETCH" $OUT/c3_traffic.txt
fi
if has soak; then
echo "== soak 60 s"; timeout 300 python tools/soak.py 60 3 > $OUT/soak.txt 2>&1; tail -2 $OUT/soak.txt
echo "== soak 60 s under the jitter build"; FA_HIP_LIB=$PWD/$L/libfa_hip_jitter.so timeout 300 python tools/soak.py 60 17 > $OUT/soak_jitter.txt 2>&1; tail -2 $OUT/soak_jitter.txt
echo "== soak 45 s, every device variant of the library"; timeout 300 python tools/soak.py 45 7 all > $OUT/soak_all.txt 2>&1; tail -2 $OUT/soak_all.txt
echo "== soak 45 s + 45 s under the jitter build: the ring forms of (128, 64, 4)+buffer, plain and speculative"; timeout 300 python tools/soak.py 45 9 ring > $OUT/soak_ring.txt 2>&1; FA_HIP_LIB=$PWD/$L/libfa_hip_jitter.so timeout 300 python tools/soak.py 45 11 ring >> $OUT/soak_ring.txt 2>&1; grep "^soak\|FAIL" $OUT/soak_ring.txt | tail -4
echo "== soak, many items per workgroup"; timeout 300 python tools/soak_many_items.py 30 5 > $OUT/soak_many_items.txt 2>&1; tail -2 $OUT/soak_many_items.txt
echo "== jitter_check (product, then jitter build)"; timeout 300 python tools/jitter_check.py > $OUT/jitter_check.txt 2>&1; FA_HIP_LIB=$PWD/$L/libfa_hip_jitter.so timeout 300 python tools/jitter_check.py >> $OUT/jitter_check.txt 2>&1; grep -c "repeat=1" $OUT/jitter_check.txt
fi
echo "== done"
