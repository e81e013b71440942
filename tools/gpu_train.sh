#!/bin/bash
# Training-step tuning session: layout / precision variants, eager and HIP-graph replay.  Output: gpurun_out/$1/
set -u
TAG=${1:-t1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_dropin.py -m gpu -x -q -k "training" ) > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
( timeout 600 python bench.py --workload train --steps 40 --warmup 5 ) > $OUT/bench_train.json 2> $OUT/bench_train.err; cat $OUT/bench_train.json; tail -3 $OUT/bench_train.err
ARCH=resnet-50 B=64 timeout 600 python tools/train_variants.py nhwc_bf16 nhwc_fp32 nchw_fp32 nchw_bf16 > $OUT/variants_rn50.log 2>&1
cat $OUT/variants_rn50.log
for v in nhwc_bf16 nchw_fp32; do ARCH=resnet-50 B=64 timeout 300 python tools/graph_variants.py $v 2>&1 | tail -3; done > $OUT/graphs_rn50.log 2>&1
cat $OUT/graphs_rn50.log
