#!/bin/bash
# training-step variants of round 2: graph replay for bf16 backbones
OUT=gpurun_out/${1:-tr2}; mkdir -p $OUT
for cfg in "resnet-110-fc 128 bf16 nhwc" "resnet-110-fc 128 bf16 nchw" "resnet-50 64 bf16 nhwc"; do
  set -- $cfg
  for graphs in 1; do
    echo "== $1 batch $2 dtype $3 layout $4 graphs=$graphs" | tee -a $OUT/train.log
    SE_TRAIN_DTYPE=$3 SE_TRAIN_LAYOUT=$4 SE_TRAIN_GRAPHS=$graphs timeout 600 python bench.py --workload train --arch $1 --batch $2 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | grep -v Warning | tail -3 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print({k: d[k] for k in ('value', 'ms_per_step', 'dtype', 'mean_loss')}, d['config']['step'], d['roofline']['gpu_busy_frac'])
    else: print(l.rstrip()[:400])
" | tee -a $OUT/train.log
  done
done
