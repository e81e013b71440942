#!/bin/bash
set -u
OUT=gpurun_out/s2; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_loss.py tests/test_abi.py -q -x ) 2>&1 | tail -12
