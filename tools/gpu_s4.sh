#!/bin/bash
export TMPDIR=/tmp
SE_PD_PROFILE=1 timeout 300 python tools/bench_kernels.py pdist --reps 1 2>&1 | grep "profile" | head -1
SE_PD_ABLATE=4 SE_PD_PROFILE=1 timeout 300 python tools/bench_kernels.py pdist --reps 1 2>&1 | grep "profile" | head -1
