#!/bin/bash
# pdist / fused iteration: retrieval tests, kernel timings.   usage: gpu_pd.sh TAG
set -u
TAG=${1:-x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
TUNING=$PWD/semantic-embeddings_amd/sehip/libsehip_tuning.so
( timeout 1200 python -m pytest tests/test_gpu_retrieval.py tests/test_gpu_topk.py -x -q ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
( timeout 300 python tools/bench_kernels.py pdist --reps 7 ) > $OUT/pdist.log 2>&1; cat $OUT/pdist.log
( timeout 300 python tools/bench_kernels.py fused --reps 5 ) > $OUT/fused.log 2>&1
( timeout 600 python tools/bench_kernels.py shard --reps 4 ) >> $OUT/fused.log 2>&1
cat $OUT/fused.log
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o fused -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py fused --reps 3 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB "python tools/bench_kernels.py fused --reps 3" | head -14 > $OUT/prof_fused_summary.txt && cat $OUT/prof_fused_summary.txt
for ab in 1; do echo "SE_PD_ABLATE=$ab"; SEHIP_LIB=$TUNING SE_PD_ABLATE=$ab timeout 300 python tools/bench_kernels.py pdist --reps 5 2>&1 | grep -v amdgpu.ids; done > $OUT/ablate.log 2>&1; cat $OUT/ablate.log
rm -rf $OUT/prof
