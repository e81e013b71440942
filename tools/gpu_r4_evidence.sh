#!/bin/bash
# Round-4 evidence session (final state of the round): full GPU test suite, bench line, kernel microbenchmarks, rocprofv3 kernel stats of
# the bench command, two-stream stress of every entry point.  Output: gpurun_out/r4ev/
set -u
OUT=gpurun_out/r4ev; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
( timeout 1200 python bench.py --steps 5 --warmup 1 ) > $OUT/bench.json 2> $OUT/bench.err; head -c 400 $OUT/bench.json; echo
for what in pdist rank fused shard hprec rownorm; do timeout 400 python tools/bench_kernels.py $what 2>&1 | grep -v amdgpu.ids; done > $OUT/kernels.log 2>&1; cat $OUT/kernels.log
( timeout 600 python tools/stress_streams.py 30 ) 2>&1 | grep -v amdgpu.ids | tee $OUT/stress_streams.log | tail -4
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r4 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train" > $OUT/prof_summary.txt && head -30 $OUT/prof_summary.txt | cut -c1-150
rm -rf $OUT/prof
