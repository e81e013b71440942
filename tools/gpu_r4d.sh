#!/bin/bash
set -u
OUT=gpurun_out/r4d; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_topk.py -x -q -s ) > $OUT/pytest_topk.log 2>&1; grep -v amdgpu.ids $OUT/pytest_topk.log | tail -6
for what in fused shard; do timeout 600 python tools/bench_kernels.py $what --reps 4 2>&1 | grep -v amdgpu.ids; done > $OUT/kernels.log 2>&1; cat $OUT/kernels.log
for V in "" _spread; do
  [ -f semantic-embeddings_amd/sehip/variants/libsehip$V.so ] || continue
  echo "== variant $V"; SEHIP_LIB=semantic-embeddings_amd/sehip/variants/libsehip$V.so timeout 300 python tools/bench_kernels.py shard --reps 2 2>&1 | grep "one chain"
  SEHIP_LIB=semantic-embeddings_amd/sehip/variants/libsehip$V.so timeout 300 python tools/bench_kernels.py fused --reps 2 2>&1 | grep "cosine"
done
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof2 -o r4d2 -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py fused --reps 4 > $GRAFT_REPO_ROOT/$OUT/prof_fused.log 2>&1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r4d -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py shard --reps 2 > $GRAFT_REPO_ROOT/$OUT/prof_shard.log 2>&1
cd $GRAFT_REPO_ROOT
for P in prof prof2; do DB=$(find $OUT/$P -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py $DB "bench_kernels.py ($P)" > $OUT/${P}_summary.txt && sed -n 6,12p $OUT/${P}_summary.txt | cut -c1-120; rm -rf $OUT/$P; done
