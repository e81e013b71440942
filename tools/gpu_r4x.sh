#!/bin/bash
set -u
OUT=gpurun_out/r4x; mkdir -p $OUT; export TMPDIR=/tmp
echo "== product"; timeout 300 python tools/bench_kernels.py hprec --hp-mode whole --reps 7 2>&1 | grep "hprec whole" | tee $OUT/hprec.log
echo "== SE_HP_SPREAD=1"; SEHIP_LIB=semantic-embeddings_amd/sehip/variants/libsehip_hpspread.so timeout 300 python tools/bench_kernels.py hprec --hp-mode whole --reps 7 2>&1 | grep "hprec whole" | tee -a $OUT/hprec.log
SEHIP_LIB=semantic-embeddings_amd/sehip/variants/libsehip_hpspread.so timeout 300 python tools/fuzz_hprec.py --seconds 20 --seed 5 2>&1 | tail -1 | tee -a $OUT/hprec.log
