#!/bin/bash
set -u
OUT=gpurun_out/r4f; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_dropin.py -x -q -k "hierarch or precision or hprec or metrics or evaluate or cli" ) 2>&1 | tail -5
timeout 300 python tools/bench_kernels.py hprec 2>&1 | grep -v amdgpu.ids | tee $OUT/hprec.log
timeout 600 python tools/fuzz_hprec.py 2>&1 | tail -4 | tee $OUT/fuzz.log
