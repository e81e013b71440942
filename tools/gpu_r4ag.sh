#!/bin/bash
set -u
OUT=gpurun_out/r4ag; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2; do ( timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu_$i.log 2>&1; tail -1 $OUT/pytest_gpu_$i.log; done
