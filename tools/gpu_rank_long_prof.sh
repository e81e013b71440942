#!/bin/bash
# per-kernel times of the sorted-runs + merge path (rows above 53,248 columns)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/rl; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o rl -- python $R/tools/bench_rank_long.py time > $OUT/rl.log 2>&1
grep rank_rows $OUT/rl.log
cd $R
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB "python tools/bench_rank_long.py time" > $OUT/summary.txt && head -30 $OUT/summary.txt | cut -c1-230
rm -rf $OUT/prof
