#!/bin/bash
set -u
OUT=gpurun_out/r4u; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_topk.py -x -q ) > $OUT/pytest_topk.log 2>&1; grep -v amdgpu.ids $OUT/pytest_topk.log | tail -4
timeout 600 python tools/bench_kernels.py shard --reps 3 2>&1 | grep -v amdgpu.ids | grep "shard retrieve" | tee $OUT/kernels.log
T=semantic-embeddings_amd/sehip/libsehip_tuning.so
SEHIP_LIB=$T SE_PF_PROFILE=1 timeout 300 python tools/bench_kernels.py shard --reps 1 2>&1 | grep -E "pf_big_kernel profile" | head -1 | tee $OUT/profile.log
for B in 0 1; do echo "== SE_PF_BIG=$B"; SEHIP_LIB=$T SE_PF_BIG=$B timeout 300 python tools/bench_kernels.py fused --reps 5 2>&1 | grep "fused retrieve"; done | tee $OUT/fused_big.log
SEHIP_LIB=$T SE_PF_BIG=1 SE_PF_PROFILE=1 timeout 300 python tools/bench_kernels.py fused --reps 1 2>&1 | grep -E "pf_big_kernel profile" | head -1 | tee -a $OUT/fused_big.log
