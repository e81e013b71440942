#!/bin/bash
# round 3, session a: fused top-k -- tests, timings, kernel breakdown
set -u
OUT=gpurun_out/r3a
mkdir -p $OUT
export TMPDIR=/tmp
TUNING=$PWD/semantic-embeddings_amd/sehip/libsehip_tuning.so
( time timeout 1200 python -m pytest tests/test_gpu_topk.py -x -q ) > $OUT/pytest_topk.log 2>&1
tail -15 $OUT/pytest_topk.log
( timeout 300 python tools/bench_kernels.py fused --reps 5 ) > $OUT/fused.log 2>&1
( SEHIP_LIB=$TUNING SE_TOPK_VERBOSE=1 timeout 300 python tools/bench_kernels.py fused --reps 1 ) >> $OUT/fused.log 2>&1
( timeout 600 python tools/bench_kernels.py shard --reps 4 ) >> $OUT/fused.log 2>&1
( SEHIP_LIB=$TUNING SE_TOPK_VERBOSE=1 timeout 600 python tools/bench_kernels.py shard --reps 2 --q 8192 ) >> $OUT/fused.log 2>&1
cat $OUT/fused.log
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o fused -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py fused --reps 3 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB "python tools/bench_kernels.py fused --reps 3" > $OUT/prof_fused_summary.txt && cat $OUT/prof_fused_summary.txt
find $OUT/prof -name "*.db" -size +20M -delete
( time timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_topk.py ) > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
