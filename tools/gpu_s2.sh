#!/bin/bash
set -u
OUT=gpurun_out/s2; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_dropin.py -q -x -k hierarchical ) 2>&1 | tail -3
timeout 300 python tools/bench_kernels.py hprec 2>&1 | grep -v amdgpu | tee $OUT/hprec.log
