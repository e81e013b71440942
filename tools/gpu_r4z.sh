#!/bin/bash
set -u
OUT=gpurun_out/r4z; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_topk.py -x -q ) > $OUT/pytest_topk.log 2>&1; grep -v amdgpu.ids $OUT/pytest_topk.log | tail -4
T=semantic-embeddings_amd/sehip/libsehip_tuning.so
for J in 28 24 20 15 1; do echo "== SE_TOPK_J=$J"; SEHIP_LIB=$T SE_TOPK_J=$J timeout 300 python tools/bench_kernels.py fused --reps 3 2>&1 | grep "fused retrieve_topk cosine"; SEHIP_LIB=$T SE_TOPK_J=$J SE_TOPK_VERBOSE=1 timeout 300 python tools/bench_kernels.py fused --reps 1 2>&1 | grep -o "redo=[0-9]* mean_candidates=[0-9.]*" | sort | uniq -c | head -1; done | tee $OUT/j.log
