#!/bin/bash
# Round-4 session L: refine kernel with LDS-staged candidate rows -- top-k parity tests, fused microbench, kernel trace.
set -u
OUT=gpurun_out/r4l; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_topk.py tests/test_gpu_dropin.py -x -q -k "topk or prefilter or sharded" ) > $OUT/pytest_topk.log 2>&1; grep -v amdgpu.ids $OUT/pytest_topk.log | tail -6
for what in fused shard; do timeout 600 python tools/bench_kernels.py $what --reps 4 2>&1 | grep -v amdgpu.ids; done > $OUT/kernels.log 2>&1; cat $OUT/kernels.log
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof2 -o r4l2 -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py fused --reps 4 > $GRAFT_REPO_ROOT/$OUT/prof_fused.log 2>&1
cd $GRAFT_REPO_ROOT
for P in prof2; do DB=$(find $OUT/$P -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py $DB "bench_kernels.py fused ($P)" > $OUT/${P}_summary.txt && sed -n 6,14p $OUT/${P}_summary.txt | cut -c1-120; rm -rf $OUT/$P; done
