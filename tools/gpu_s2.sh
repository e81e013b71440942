#!/bin/bash
set -u
OUT=gpurun_out/s2; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -k "rank or golden or full_size" ) > $OUT/pytest_rank.log 2>&1; tail -15 $OUT/pytest_rank.log
timeout 300 python tools/bench_kernels.py rank > $OUT/rank.log 2>&1; cat $OUT/rank.log
SE_RR_PROFILE=1 timeout 300 python tools/bench_kernels.py rank --reps 1 >> $OUT/rank.log 2>&1; tail -3 $OUT/rank.log
timeout 300 python tools/bench_kernels.py rank --n 10000 >> $OUT/rank.log 2>&1; tail -1 $OUT/rank.log
