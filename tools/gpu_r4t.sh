#!/bin/bash
set -u
OUT=gpurun_out/r4t; mkdir -p $OUT; export TMPDIR=/tmp
T=semantic-embeddings_amd/sehip/libsehip_tuning.so
for B in 0 1; do echo "== SE_PF_BIG=$B"; SEHIP_LIB=$T SE_PF_BIG=$B timeout 300 python tools/bench_kernels.py fused --reps 5 2>&1 | grep "fused retrieve"; done | tee $OUT/fused_big.log
SEHIP_LIB=$T SE_PF_BIG=1 SE_PF_PROFILE=1 timeout 300 python tools/bench_kernels.py fused --reps 1 2>&1 | grep -E "pf_big_kernel profile" | head -2 | tee -a $OUT/fused_big.log
