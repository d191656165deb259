#!/bin/bash
# Autotune sweep = KERNELS=tune (+ the CDNA4-native shapes) through the profiler table
# (reference: tools/benchmark/benchmark_autotune.sh -> KERNELS=tune ncu_bench.py).
cd "$(dirname "$0")"
KERNELS=${KERNELS:-tune} python rocprof_bench.py --seq_lens "${1:-4096}" --pmc
KERNELS=native python rocprof_bench.py --seq_lens "${1:-4096}" --pmc
