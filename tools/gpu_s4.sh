#!/bin/bash
set -u
OUT=gpurun_out/s4; mkdir -p $OUT; export TMPDIR=/tmp
for lib in libsehip.so libsehip_bk32_w3.so libsehip_bk32_w4.so; do
  echo "== $lib"
  export SEHIP_LIB=$PWD/semantic-embeddings_amd/sehip/$lib
  timeout 600 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -x -k "pairwise or golden or full_size" 2>&1 | tail -1
  for ab in 0 1; do SE_PD_ABLATE=$ab timeout 300 python tools/bench_kernels.py pdist 2>&1 | grep "pdist"; done
  SE_PD_PROFILE=1 timeout 300 python tools/bench_kernels.py pdist --reps 1 2>&1 | grep "profile" | head -1
done | tee $OUT/pdist_variants.log
