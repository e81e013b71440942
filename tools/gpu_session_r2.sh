#!/bin/bash
# Round-2 GPU session: GPU tests, bench line (cosine = headline, Euclidean = the CLI default branch), kernel microbenchmarks.
#   tools/gpu_session_r2.sh <tag>
set -u
TAG=${1:-r2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/host.txt; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/host.txt
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.log 2>&1
tail -6 $OUT/pytest_gpu.log
( timeout 900 python bench.py --steps 5 --warmup 1 ) > $OUT/bench.json 2> $OUT/bench.err
cut -c1-1500 $OUT/bench.json; tail -3 $OUT/bench.err
( timeout 600 python bench.py --steps 5 --warmup 1 --metric euclid --no-train --no-sharded --no-cpu-baseline ) > $OUT/bench_euclid.json 2>> $OUT/bench.err
cut -c1-700 $OUT/bench_euclid.json
for what in pdist rank topk loss hprec shard; do timeout 300 python tools/bench_kernels.py $what; done > $OUT/kernels.log 2>&1
grep -v Warning $OUT/kernels.log
python __graft_entry__.py smoke 2>&1 | tail -2
