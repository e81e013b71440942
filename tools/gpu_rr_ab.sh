#!/bin/bash
# A/B of the ranking-kernel build variants: correctness (rank tests of the GPU suite) + microbenchmark per variant library
set -u
OUT=gpurun_out/${1:-rrab}
mkdir -p $OUT
for v in default; do
  if [ "$v" = "default" ]; then unset SEHIP_LIB; else export SEHIP_LIB=$PWD/semantic-embeddings_amd/sehip/libsehip_rr_$v.so; fi
  echo "=== variant $v" | tee -a $OUT/ab.log
  ( timeout 600 python -m pytest tests/test_gpu_retrieval.py -q -x -k "rank or full_size or benchmarked or golden" 2>&1 | tail -3 ) | tee -a $OUT/ab.log
  ( timeout 300 python tools/bench_kernels.py rank --reps 7 2>&1 | grep -v Warning ) | tee -a $OUT/ab.log
done
