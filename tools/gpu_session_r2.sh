#!/bin/bash
# Round-2 GPU session: GPU tests, bench line, rocprof kernel summary.   tools/gpu_session_r2.sh <tag> [quick]
set -u
TAG=${1:-r2a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/host.txt; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/host.txt
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.log 2>&1
tail -15 $OUT/pytest_gpu.log
( timeout 900 python bench.py --steps 5 --warmup 1 ) > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json; tail -5 $OUT/bench.err
