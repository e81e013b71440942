#!/bin/bash
# Round-4 session P: 256 x 256 filter kernel -- top-k parity tests, shard / fused microbench, phase profile.
set -u
OUT=gpurun_out/r4p; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_topk.py -x -q ) > $OUT/pytest_topk.log 2>&1; grep -v amdgpu.ids $OUT/pytest_topk.log | tail -12
for what in shard fused; do timeout 600 python tools/bench_kernels.py $what --reps 3 2>&1 | grep -v amdgpu.ids; done > $OUT/kernels.log 2>&1; cat $OUT/kernels.log
T=semantic-embeddings_amd/sehip/libsehip_tuning.so
SEHIP_LIB=$T SE_PF_PROFILE=1 SE_TOPK_VERBOSE=1 timeout 300 python tools/bench_kernels.py shard --reps 1 2>&1 | grep -E "profile\]|prefilter:" | sort | uniq -c | sort -rn | head -8 | tee $OUT/profile.log
SEHIP_LIB=$T SE_PF_BIG=0 timeout 300 python tools/bench_kernels.py shard --reps 2 2>&1 | grep -E "shard retrieve" | sed 's/^/[SE_PF_BIG=0] /' | tee -a $OUT/profile.log
SEHIP_LIB=$T SE_PF_BIG=1 timeout 300 python tools/bench_kernels.py shard --reps 2 2>&1 | grep -E "shard retrieve" | sed 's/^/[SE_PF_BIG=1] /' | tee -a $OUT/profile.log
