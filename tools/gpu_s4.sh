#!/bin/bash
set -u
OUT=gpurun_out/s4; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -x -k "pairwise or golden or full_size" 2>&1 | tail -3
for ab in 0 4 1; do echo "ABLATE=$ab"; SE_PD_ABLATE=$ab timeout 300 python tools/bench_kernels.py pdist 2>&1 | grep pdist; done | tee $OUT/pdist_ablate3.log
timeout 300 python tools/bench_kernels.py pdist --d 200 2>&1 | grep pdist
timeout 300 python tools/bench_kernels.py pdist --d 1000 --n 20000 2>&1 | grep pdist
