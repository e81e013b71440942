#!/bin/bash
# pre-filter: parity tests + microbenchmarks + phase profiles (s_memtime, tuning build)
set -u
OUT=gpurun_out/r4c; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_topk.py -x -q -s ) > $OUT/pytest_topk.log 2>&1; grep -v amdgpu.ids $OUT/pytest_topk.log | tail -12
for what in fused shard; do timeout 600 python tools/bench_kernels.py $what --reps 4 2>&1 | grep -v amdgpu.ids; done > $OUT/kernels.log 2>&1; cat $OUT/kernels.log
export SEHIP_LIB=semantic-embeddings_amd/sehip/libsehip_tuning.so
echo "== profiles"; SE_PF_PROFILE=1 timeout 300 python tools/bench_kernels.py shard --reps 2 2>&1 | grep "profile\]" | grep "epi=2" | tail -1 | tee $OUT/profile_shard.log
SE_PF_PROFILE=1 timeout 300 python tools/bench_kernels.py fused --reps 2 2>&1 | grep "profile\]" | grep "epi=2" | sort -u -k3,3 | tee $OUT/profile_fused.log
SE_TOPK_VERBOSE=1 timeout 300 python tools/bench_kernels.py fused --reps 1 2>&1 | grep "prefilter:" | sort | uniq -c | tee $OUT/fused_stats.log
SE_TOPK_VERBOSE=1 timeout 300 python tools/bench_kernels.py shard --reps 1 2>&1 | grep "prefilter:" | sort | uniq -c | tee $OUT/shard_stats.log
unset SEHIP_LIB
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r4c -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py shard --reps 2 > $GRAFT_REPO_ROOT/$OUT/prof_shard.log 2>&1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof2 -o r4c2 -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py fused --reps 4 > $GRAFT_REPO_ROOT/$OUT/prof_fused.log 2>&1
cd $GRAFT_REPO_ROOT
for P in prof prof2; do DB=$(find $OUT/$P -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py $DB "bench_kernels.py ($P)" > $OUT/${P}_summary.txt && sed -n 6,16p $OUT/${P}_summary.txt | cut -c1-150; rm -rf $OUT/$P; done
