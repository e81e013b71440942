#!/bin/bash
set -u
OUT=gpurun_out/s7; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -x -k "rank or golden or full_size" ) 2>&1 | tail -2
for m in cosine euclid; do timeout 300 python bench.py --steps 5 --warmup 1 --metric $m --no-train --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m step ms', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})"; done | tee $OUT/rank_peel.log
