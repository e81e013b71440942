#!/bin/bash
# round-5 GPU session runner: tools/gpu_r5.sh <stage> ; logs under gpurun_out/r5_<stage>.log
stage=${1:-img}
mkdir -p gpurun_out
export SEHIP_LIB=${SEHIP_LIB:-$PWD/semantic-embeddings_amd/sehip/variants/libsehip_dev.so}
log=gpurun_out/r5_$stage.log
: > $log
case $stage in
img)
  echo "== check, forced image path" >> $log
  SE_RANK_PEEL=3 timeout 600 python tools/dev_img.py check >> $log 2>&1
  echo "== check, detector" >> $log
  timeout 600 python tools/dev_img.py check >> $log 2>&1
  echo "== time, detector" >> $log
  SE_RANK_VERBOSE=1 timeout 300 python tools/dev_img.py time >> $log 2>&1
  echo "== time, forced plain 3-pass" >> $log
  SE_RANK_PEEL=0 timeout 300 python tools/dev_img.py time >> $log 2>&1
  echo "== profile, forced image" >> $log
  SE_RANK_PEEL=3 SE_RR_PROFILE=1 timeout 300 python tools/dev_img.py time --reps 1 >> $log 2>&1
  echo "== profile, forced window two-pass on Euclid (for comparison)" >> $log
  SE_RANK_PEEL=2 SE_RR_PROFILE=1 timeout 300 python tools/dev_img.py time --reps 1 >> $log 2>&1
  ;;
pd)
  unset SEHIP_LIB
  timeout 1200 python -m pytest tests/test_gpu_retrieval.py -x -q -m gpu -k "pairwise or golden or benchmarked or full_size" >> $log 2>&1
  timeout 600 python tools/bench_kernels.py pdist --reps 7 >> $log 2>&1
  ;;
topk)
  unset SEHIP_LIB
  timeout 1800 python -m pytest tests/test_gpu_topk.py tests/test_gpu_loss.py -x -q -m gpu >> $log 2>&1
  timeout 1800 python -m pytest tests/test_gpu_retrieval.py tests/test_gpu_dropin.py -x -q -m gpu -k "topk or larger or sharded or retrieve or row_norms or row_sqnorm" >> $log 2>&1
  timeout 300 python tools/fuzz_topk.py --seconds 120 >> $log 2>&1
  timeout 600 python tools/bench_kernels.py fused >> $log 2>&1
  timeout 600 python tools/bench_kernels.py shard >> $log 2>&1
  timeout 600 python tools/bench_kernels.py rownorm >> $log 2>&1
  ;;
skew)
  unset SEHIP_LIB
  timeout 900 python tools/topk_skew.py >> $log 2>&1
  timeout 900 python tools/topk_skew.py --n 160146 --q 20000 --d 1000 --classes 125 >> $log 2>&1
  ;;
alltests)
  unset SEHIP_LIB
  timeout 3400 python -m pytest tests -x -q -m gpu >> $log 2>&1
  ;;
tests)
  unset SEHIP_LIB
  timeout 2400 python -m pytest tests/test_gpu_retrieval.py -x -q -m gpu -k "rank or full_size or benchmarked or golden" >> $log 2>&1
  echo "== fuzz_rank" >> $log
  timeout 400 python tools/fuzz_rank.py --seconds 150 --seed 5 >> $log 2>&1
  echo "== bench" >> $log
  timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_line.json 2>> $log
  cat gpurun_out/r5_bench_line.json >> $log
  ;;
esac
tail -60 $log
