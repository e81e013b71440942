#!/bin/bash
set -u
OUT=gpurun_out/r4w; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_topk.py tests/test_gpu_retrieval.py tests/test_gpu_dropin.py -x -q -k "merge or sharded or two_process" ) > $OUT/pytest_merge.log 2>&1; grep -v amdgpu.ids $OUT/pytest_merge.log | tail -5
timeout 600 python tools/bench_kernels.py shard --reps 3 2>&1 | grep -v amdgpu.ids | grep "merge" | tee $OUT/kernels.log
