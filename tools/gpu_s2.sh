#!/bin/bash
set -u
OUT=gpurun_out/s2; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -x -k rank ) 2>&1 | tail -3
SE_RR_PROFILE=1 timeout 300 python tools/bench_kernels.py rank --reps 1 2>&1 | grep -v amdgpu.ids | tee $OUT/rank_prof.log
timeout 300 python tools/bench_kernels.py rank 2>&1 | grep "rank q" | tee -a $OUT/rank_prof.log
SE_RANK_SAFE=1 timeout 300 python tools/bench_kernels.py rank 2>&1 | grep "rank q" | tee -a $OUT/rank_prof.log
