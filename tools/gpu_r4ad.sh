#!/bin/bash
set -u
OUT=gpurun_out/r4ad; mkdir -p $OUT; export TMPDIR=/tmp
for what in fused shard; do timeout 400 python tools/bench_kernels.py $what --reps 7 2>&1 | grep -v amdgpu.ids | grep retrieve; done | tee $OUT/kernels.log
( time timeout 1200 python -m pytest tests/test_gpu_topk.py -x -q ) > $OUT/pytest_topk.log 2>&1; grep -E "passed|failed" $OUT/pytest_topk.log | tail -2
