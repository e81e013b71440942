#!/bin/bash
# Round-4 session B: the bf16 pre-filter of se_retrieve_topk -- parity tests, error-bound test, microbenchmarks, statistics.
set -u
OUT=gpurun_out/r4b; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_topk.py -x -q -s ) > $OUT/pytest_topk.log 2>&1; grep -v amdgpu.ids $OUT/pytest_topk.log | tail -30
for what in fused shard; do timeout 600 python tools/bench_kernels.py $what --reps 4 2>&1 | grep -v amdgpu.ids; done > $OUT/kernels.log 2>&1; cat $OUT/kernels.log
SEHIP_LIB=semantic-embeddings_amd/sehip/libsehip_tuning.so SE_TOPK_VERBOSE=1 timeout 600 python tools/bench_kernels.py shard --reps 2 2>&1 | grep "prefilter:" | sort | uniq -c | head -8 > $OUT/shard_stats.log; cat $OUT/shard_stats.log
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r4b -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py shard --reps 2 > $GRAFT_REPO_ROOT/$OUT/prof_shard.log 2>&1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof2 -o r4b2 -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py fused --reps 4 > $GRAFT_REPO_ROOT/$OUT/prof_fused.log 2>&1
cd $GRAFT_REPO_ROOT
for P in prof prof2; do DB=$(find $OUT/$P -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py $DB "bench_kernels.py ($P)" > $OUT/${P}_summary.txt && head -24 $OUT/${P}_summary.txt; rm -rf $OUT/$P; done
