#!/bin/bash
set -u
OUT=gpurun_out/r4n; mkdir -p $OUT; export TMPDIR=/tmp
T=semantic-embeddings_amd/sehip/libsehip_tuning.so
for S in 2048 4096; do for R in 1 0; do
  echo "== SE_TOPK_SAMPLES=$S SE_RF_ROWS=$R"
  SEHIP_LIB=$T SE_RF_ROWS=$R SE_TOPK_SAMPLES=$S timeout 300 python tools/bench_kernels.py fused --reps 5 2>&1 | grep -E "fused retrieve"
done; done 2>&1 | tee $OUT/samples.log
SEHIP_LIB=$T SE_TOPK_SAMPLES=4096 SE_TOPK_VERBOSE=1 timeout 300 python tools/bench_kernels.py fused --reps 1 2>&1 | grep -E "prefilter:" | sort | uniq -c | tee -a $OUT/samples.log
