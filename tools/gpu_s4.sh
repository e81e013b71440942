#!/bin/bash
set -u
OUT=gpurun_out/s4; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -x -k "pairwise or golden or full_size" 2>&1 | tail -3
for ab in 0 1; do echo "ABLATE=$ab"; SE_PD_ABLATE=$ab timeout 300 python tools/bench_kernels.py pdist 2>&1 | grep "pdist"; done | tee $OUT/pdist_ablate5.log
SE_PD_PROFILE=1 timeout 300 python tools/bench_kernels.py pdist --reps 1 2>&1 | grep "profile" | head -2
