#!/bin/bash
set -u
OUT=gpurun_out/r4y; mkdir -p $OUT; export TMPDIR=/tmp
T=semantic-embeddings_amd/sehip/libsehip_tuning.so
for J in 28 24 20 17 15; do echo "== SE_TOPK_J=$J"; SEHIP_LIB=$T SE_TOPK_J=$J timeout 300 python tools/bench_kernels.py fused --reps 5 2>&1 | grep "fused retrieve_topk cosine"; SEHIP_LIB=$T SE_TOPK_J=$J SE_TOPK_VERBOSE=1 timeout 300 python tools/bench_kernels.py fused --reps 1 2>&1 | grep -o "redo=[0-9]* mean_candidates=[0-9.]*" | sort | uniq -c | head -2; done | tee $OUT/j.log
