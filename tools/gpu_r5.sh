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
train)
  unset SEHIP_LIB
  for cfg in "" "PYTORCH_MIOPEN_SUGGEST_NHWC_BATCHNORM=1" "PYTORCH_MIOPEN_SUGGEST_NHWC_BATCHNORM=1 PYTORCH_MIOPEN_SUGGEST_NHWC=1"; do
    echo "== resnet-50 b128 bf16 [$cfg]" >> $log
    env $cfg timeout 600 python bench.py --workload train --arch resnet-50 --batch 128 --steps 20 --warmup 5 --no-cpu-baseline 2>>$log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['backbone'])" >> $log 2>&1
  done
  for b in 128 256 512; do
    echo "== resnet-110-fc fp32 graphs batch $b" >> $log
    timeout 600 python bench.py --workload train --arch resnet-110-fc --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>>$log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['step'], d['roofline']['frac'])" >> $log 2>&1
  done
  ;;
skew)
  unset SEHIP_LIB
  timeout 900 python tools/topk_skew.py >> $log 2>&1
  timeout 900 python tools/topk_skew.py --n 160146 --q 20000 --d 1000 --classes 125 >> $log 2>&1
  ;;
two)
  echo "== two-pass tests through the dev library (forced 2 and detector)" >> $log
  timeout 900 python -m pytest tests/test_gpu_retrieval.py -x -q -m gpu -k "two_pass or skewed or special or boundaries or matches_canonical" >> $log 2>&1
  SE_RANK_PEEL=2 timeout 600 python tools/dev_img.py check >> $log 2>&1
  timeout 300 python tools/fuzz_rank.py --seconds 100 --seed 11 >> $log 2>&1
  timeout 300 python tools/dev_img.py time >> $log 2>&1
  ;;
alltests)
  unset SEHIP_LIB
  timeout 3400 python -m pytest tests -x -q -m gpu >> $log 2>&1
  ;;
fuzz)
  unset SEHIP_LIB
  timeout 400 python tools/fuzz_rank.py --seconds 150 --seed 21 >> $log 2>&1
  timeout 400 python tools/fuzz_rank.py --seconds 60 --seed 22 --long >> $log 2>&1
  timeout 400 python tools/fuzz_topk.py --seconds 100 >> $log 2>&1
  timeout 400 python tools/fuzz_hprec.py --seconds 40 >> $log 2>&1
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
