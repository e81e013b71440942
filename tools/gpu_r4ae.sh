#!/bin/bash
set -u
OUT=gpurun_out/r4ae; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof2 -o r4ae -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py fused --reps 4 > $GRAFT_REPO_ROOT/$OUT/prof_fused.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/prof2 -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py $DB "bench_kernels.py fused --reps 4" > $OUT/prof_fused_summary.txt && sed -n 1,22p $OUT/prof_fused_summary.txt | cut -c1-150; rm -rf $OUT/prof2
