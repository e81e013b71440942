#!/bin/bash
# One GPU-box session: correctness, bench, profile, tuning probes.  Output under gpurun_out/$1/
set -u
TAG=${1:-s1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/host.txt; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/host.txt
python -c "import numpy; numpy.show_config()" 2>&1 | grep -iE "name|openblas|version" >> $OUT/host.txt
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
( timeout 900 python bench.py --steps 5 --warmup 1 ) > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
for what in pdist rank topk loss; do timeout 300 python tools/bench_kernels.py $what; done > $OUT/kernels.log 2>&1
cat $OUT/kernels.log
for ab in 1 2 3; do echo "SE_PD_ABLATE=$ab"; SE_PD_ABLATE=$ab timeout 300 python tools/bench_kernels.py pdist; done > $OUT/ablate.log 2>&1
SE_PD_NOSTAGGER=1 timeout 300 python tools/bench_kernels.py pdist >> $OUT/ablate.log 2>&1
cat $OUT/ablate.log
hipcc --offload-arch=gfx950 -O3 tools/probes/store_patterns.hip -o /tmp/store_patterns && timeout 120 /tmp/store_patterns > $OUT/store_patterns.log 2>&1
cat $OUT/store_patterns.log
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train > $OUT/prof.log 2>&1
find $OUT/prof -name "*.db" | head
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train" > $OUT/prof_summary.txt && cat $OUT/prof_summary.txt
find $OUT/prof -name "*.db" -size +20M -delete
