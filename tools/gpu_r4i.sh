#!/bin/bash
set -u
OUT=gpurun_out/r4i; mkdir -p $OUT; export TMPDIR=/tmp
for V in bk64w3 bk64w4; do
  echo "== variant $V"; SEHIP_LIB=semantic-embeddings_amd/sehip/variants/libsehip_$V.so timeout 300 python tools/bench_kernels.py shard --reps 2 2>&1 | grep "one chain"
  SEHIP_LIB=semantic-embeddings_amd/sehip/variants/libsehip_$V.so timeout 300 python tools/bench_kernels.py fused --reps 3 2>&1 | grep "cosine"
done 2>&1 | tee $OUT/bk64.log
