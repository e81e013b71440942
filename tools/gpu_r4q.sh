#!/bin/bash
set -u
OUT=gpurun_out/r4q; mkdir -p $OUT; export TMPDIR=/tmp
T=semantic-embeddings_amd/sehip/libsehip_tuning.so
SEHIP_LIB=$T SE_PF_PROFILE=1 timeout 300 python tools/bench_kernels.py shard --reps 1 2>&1 | grep -E "pf_big_kernel profile" | head -3 | tee $OUT/profile.log
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r4q -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py shard --reps 2 > $GRAFT_REPO_ROOT/$OUT/prof_shard.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/prof -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py $DB "bench_kernels.py shard" > $OUT/prof_summary.txt && sed -n 6,16p $OUT/prof_summary.txt | cut -c1-130; rm -rf $OUT/prof
