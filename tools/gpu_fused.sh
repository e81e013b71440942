#!/bin/bash
# fused top-k iteration: tests (optional), timings, per-kernel breakdown.   usage: gpu_fused.sh TAG [notest]
set -u
TAG=${1:-x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
TUNING=$PWD/semantic-embeddings_amd/sehip/libsehip_tuning.so
if [ "${2:-}" != "notest" ]; then
  ( timeout 900 python -m pytest tests/test_gpu_topk.py -x -q ) > $OUT/pytest_topk.log 2>&1; tail -3 $OUT/pytest_topk.log
fi
( timeout 300 python tools/bench_kernels.py fused --reps 5 ) > $OUT/fused.log 2>&1
( timeout 600 python tools/bench_kernels.py shard --reps 4 ) >> $OUT/fused.log 2>&1
cat $OUT/fused.log
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o fused -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py fused --reps 3 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB "python tools/bench_kernels.py fused --reps 3" | head -16 > $OUT/prof_fused_summary.txt && cat $OUT/prof_fused_summary.txt
# floor of the main pass: the same tile walk with no epilogue at all (tuning build; results are garbage, only the pdist_kernel<..,2> time counts)
cd /tmp
SEHIP_LIB=$TUNING SE_PD_ABLATE=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_ab -o ab -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py fused --reps 1 > $GRAFT_REPO_ROOT/$OUT/prof_ab.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/prof_ab -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB "SE_PD_ABLATE=1 fused" | grep "pdist_kernel" | head -4
rm -rf $OUT/prof $OUT/prof_ab
