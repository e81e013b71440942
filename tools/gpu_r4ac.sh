#!/bin/bash
set -u
OUT=gpurun_out/r4ac; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python tools/fuzz_topk.py --seconds 150 --seed 11 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/fuzz_topk.log
