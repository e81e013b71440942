#!/bin/bash
# Training-step session: GPU tests of the trainer, train bench (graph replay and eager), rocprof of the default step.  Output: gpurun_out/$1/
set -u
TAG=${1:-t1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_dropin.py -m gpu -x -q -k "training" ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
( timeout 600 python bench.py --workload train --steps 40 --warmup 5 ) > $OUT/bench_train.json 2> $OUT/bench_train.err; cat $OUT/bench_train.json; tail -3 $OUT/bench_train.err
( SE_TRAIN_GRAPHS=0 timeout 600 python bench.py --workload train --steps 40 --warmup 5 ) > $OUT/bench_train_eager.json 2>> $OUT/bench_train.err; cat $OUT/bench_train_eager.json
rocprofv3 --kernel-trace --stats -d $OUT/prof -o t -- python bench.py --workload train --steps 40 --warmup 5 > $OUT/prof.log 2>&1
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB "python bench.py --workload train --steps 40 --warmup 5" > $OUT/prof_summary.txt && head -45 $OUT/prof_summary.txt
find $OUT/prof -name "*.db" -size +20M -delete
