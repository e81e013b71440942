#!/bin/bash
# Round-2 (late) evidence: rocprofv3 kernel summary of the default bench command (retrieval + metrics legs) and the PMC passes of the
# hierarchical-precision kernel (separate runs, no tracing flags with --pmc)
set -u
OUT=gpurun_out/${1:-prof2b}; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --no-sharded"
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r2b -- $CMD > $OUT/prof.log 2>&1
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB "$CMD" > $OUT/prof_summary.txt && cat $OUT/prof_summary.txt
find $OUT/prof -name "*.db" -size +20M -delete
bash tools/gpu_profile_hprec.sh $(basename $OUT)/hp > $OUT/hprec_pmc.txt 2>&1
cat $OUT/hprec_pmc.txt
