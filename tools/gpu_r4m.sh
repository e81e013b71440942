#!/bin/bash
# Round-4 session M: phase profile of pf_refine_kernel (tuning build), staged rows vs lane-by-lane rows.
set -u
OUT=gpurun_out/r4m; mkdir -p $OUT; export TMPDIR=/tmp
T=semantic-embeddings_amd/sehip/libsehip_tuning.so
for R in 1 0; do
  echo "== SE_RF_ROWS=$R"
  SEHIP_LIB=$T SE_RF_ROWS=$R SE_TOPK_VERBOSE=1 timeout 300 python tools/bench_kernels.py fused --reps 2 2>&1 | grep -E "profile|fused retrieve|prefilter:" | sort | uniq -c | sort -rn | head -8
done 2>&1 | tee $OUT/refine_profile.log
