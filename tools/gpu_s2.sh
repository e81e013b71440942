#!/bin/bash
set -u
OUT=gpurun_out/s2; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_retrieval.py tests/test_gpu_dropin.py -q -x -k "topk or retrieve or sharded" ) 2>&1 | tail -4
timeout 300 python tools/bench_kernels.py topk 2>&1 | grep -v amdgpu | tee $OUT/topk.log
SE_TOPK_STREAM=1 timeout 300 python tools/bench_kernels.py topk 2>&1 | grep -v amdgpu | tee -a $OUT/topk.log
timeout 300 python tools/bench_kernels.py topk --k 10 2>&1 | grep -v amdgpu | tee -a $OUT/topk.log
timeout 300 python tools/bench_kernels.py topk --k 1000 2>&1 | grep -v amdgpu | tee -a $OUT/topk.log
timeout 300 python tools/bench_kernels.py topk --n 10000 2>&1 | grep -v amdgpu | tee -a $OUT/topk.log
