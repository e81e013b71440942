#!/bin/bash
# Round-4 session O: whole GPU suite (with the two-stream regression tests), every entry point under two-stream contention, fuzzers.
set -u
OUT=gpurun_out/r4o; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
( timeout 900 python tools/stress_streams.py 40 ) 2>&1 | grep -v amdgpu.ids | tee $OUT/stress_streams.log
( timeout 200 python tools/fuzz_rank.py --seconds 90 --seed 41 ) 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/fuzz_rank.log
( timeout 200 python tools/fuzz_hprec.py --seconds 60 --seed 41 ) 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/fuzz_hprec.log
