#!/bin/bash
set -u
OUT=gpurun_out/r4r; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_topk.py -x -q ) > $OUT/pytest_topk.log 2>&1; grep -v amdgpu.ids $OUT/pytest_topk.log | tail -12
timeout 600 python tools/bench_kernels.py shard --reps 3 2>&1 | grep -v amdgpu.ids | tee $OUT/kernels.log
T=semantic-embeddings_amd/sehip/libsehip_tuning.so
SEHIP_LIB=$T SE_PF_PROFILE=1 timeout 300 python tools/bench_kernels.py shard --reps 1 2>&1 | grep -E "pf_big_kernel profile" | head -2 | tee $OUT/profile.log
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r4r -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py shard --reps 2 > $GRAFT_REPO_ROOT/$OUT/prof_shard.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/prof -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py $DB "bench_kernels.py shard" > $OUT/prof_summary.txt && sed -n 6,10p $OUT/prof_summary.txt | cut -c1-130; rm -rf $OUT/prof
