#!/bin/bash
# rocprofv3 kernel statistics of the training legs (fresh kernel_mix for the bench line): ResNet-50 bf16 channels_last (batch 128) and ResNet-110-fc fp32 graph replay
set -u
OUT=gpurun_out/r4h; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/p50 -o r50 -- python $GRAFT_REPO_ROOT/bench.py --workload train --arch resnet-50 --batch 128 --steps 6 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/r50.log 2>&1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/p110 -o r110 -- python $GRAFT_REPO_ROOT/bench.py --workload train --arch resnet-110-fc --batch 128 --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/r110.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/p50 -name "*.db" | head -1); python tools/rocprof_window.py $DB 150 "bench.py --workload train --arch resnet-50 --batch 128 (bf16 autocast, channels_last, eager)" | tee $OUT/p50_window.txt | cut -c1-170
DB=$(find $OUT/p110 -name "*.db" | head -1); python tools/rocprof_window.py $DB 250 "bench.py --workload train --arch resnet-110-fc --batch 128 (fp32 NCHW, HIP-graph replay)" | tee $OUT/p110_window.txt | cut -c1-170
rm -rf $OUT/p50 $OUT/p110
tail -2 $OUT/r50.log | cut -c1-400
