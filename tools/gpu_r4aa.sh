#!/bin/bash
set -u
OUT=gpurun_out/r4aa; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_topk.py -x -q ) > $OUT/pytest_topk.log 2>&1; grep -E "passed|failed" $OUT/pytest_topk.log | tail -2
timeout 600 python tools/bench_kernels.py fused --reps 7 2>&1 | grep "fused retrieve" | tee $OUT/kernels.log
