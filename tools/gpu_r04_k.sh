#!/bin/bash
# round 4, call K: GPU tier on the library as committed (adaptive probe lock, two-thread test)
export PYTHONPATH=$PWD:$PYTHONPATH
OUT=gpurun_out/r04k; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $OUT/pytest_gpu.txt 2>&1; tail -8 $OUT/pytest_gpu.txt
