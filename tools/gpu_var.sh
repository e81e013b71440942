#!/bin/bash
# variant comparison: for each libsehip variant named on the command line (or "product"), time pdist / fused / shard
set -u
OUT=gpurun_out/var; mkdir -p $OUT
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "product" ]; then unset SEHIP_LIB; else export SEHIP_LIB=$PWD/semantic-embeddings_amd/sehip/variants/libsehip_$v.so; fi
  echo "=== $v"
  timeout 300 python tools/bench_kernels.py pdist --reps 7 2>&1 | grep -v amdgpu.ids | grep "pdist"
  timeout 300 python tools/bench_kernels.py fused --reps 5 2>&1 | grep "^fused"
  timeout 600 python tools/bench_kernels.py shard --reps 3 2>&1 | grep "^shard"
done 2>&1 | tee $OUT/variants_$(date +%H%M%S).log
