#!/bin/bash
# round-6 GPU session runner: tools/gpu_r6.sh <stage> ; logs under gpurun_out/r6_<stage>.log
# stages img / two / pmcrk use the fast-to-build development library (three instantiations of the ranking kernel):
#   make -C semantic-embeddings_amd/csrc variant NAME=dev VFLAGS=-DSE_RR_DEV
stage=${1:-img}
mkdir -p gpurun_out
export SEHIP_LIB=${SEHIP_LIB:-$PWD/semantic-embeddings_amd/sehip/variants/libsehip_dev.so}
log=gpurun_out/r6_$stage.log
: > $log
case $stage in
idx16)
  unset SEHIP_LIB
  timeout 1800 python -m pytest tests/test_gpu_retrieval.py tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -4 >> $log
  timeout 300 python tools/eval_e2e.py 2>&1 | grep -v amdgpu.ids >> $log
  timeout 300 python tools/eval_e2e.py --euclid 2>&1 | grep -v amdgpu.ids >> $log
  timeout 300 python - >> $log 2>&1 <<'PY'
import sys, os, numpy as np, torch
sys.path[:0] = [os.path.join(os.getcwd(), "semantic-embeddings_amd"), os.getcwd()]
import sehip
n = 50000
x = torch.from_numpy(np.random.default_rng(0).standard_normal((n, 100)).astype(np.float32)).cuda()
sehip.normalize_rows_(x)
pd = sehip.pairwise_dist(x, x, metric=sehip.METRIC_COSINE)
for name, kw, dt in (("int32", {}, torch.int32), ("uint16", {"idx16": True}, torch.int16)):
    rk = torch.empty((n, n), dtype=dt, device="cuda")
    sehip.rank_rows(pd, out=rk, **kw); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); sehip.rank_rows(pd, out=rk, **kw); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print("rank_rows %s ranks 50k x 50k: median %.3f ms" % (name, float(np.median(ts))))
    C = 100
    rng = np.random.default_rng(1)
    cls_h = rng.integers(0, C, size=n).astype(np.int32)
    tab = rng.random((C, C)); tab = (tab + tab.T) / 2; np.fill_diagonal(tab, 1.0)
    counts = np.bincount(cls_h, minlength=C)
    best = np.stack([np.cumsum(np.repeat(tab[c][np.argsort(-tab[c], kind="stable")], counts[np.argsort(-tab[c], kind="stable")])) for c in range(C)])
    cls = torch.from_numpy(cls_h).cuda(); tab_d, best_d = torch.from_numpy(tab).cuda(), torch.from_numpy(best).cuda()
    qidx = torch.arange(n, dtype=torch.int32, device="cuda"); ks = torch.arange(1, 251, dtype=torch.int32, device="cuda")
    curves = sehip.hprec_reciprocal_curves(best_d, best_d)
    run = lambda: sehip.hierarchical_precision(rk, cls, cls, qidx, tab_d, tab_d, best_d, best_d, ks, ahp_len=0, want_ap=True, curves=curves)
    res = run(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print("hierarchical_precision on %s ranks: median %.3f ms  checksum %.12f" % (name, float(np.median(ts)), float(res.sum())))
    del rk
PY
  ;;
topk6)
  unset SEHIP_LIB
  timeout 1800 python -m pytest tests/test_gpu_topk.py -x -q -m gpu 2>&1 | tail -3 >> $log
  timeout 300 python tools/fuzz_topk.py --seconds 90 2>&1 | tail -2 >> $log
  timeout 600 python tools/bench_kernels.py fused 2>&1 | grep -v amdgpu.ids >> $log
  ;;
hprec)
  unset SEHIP_LIB
  timeout 1200 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -3 >> $log
  timeout 300 python tools/fuzz_hprec.py --seconds 60 2>&1 | tail -2 >> $log
  timeout 600 python tools/bench_kernels.py hprec 2>&1 | grep -v amdgpu.ids >> $log
  ;;
pdab)
  # A/B of distance-kernel variants: tools/gpu_r6.sh pdab name...   ("prod" = the product library)
  shift
  for v in "$@"; do
    echo "== $v" >> $log
    if [ $v = prod ]; then unset SEHIP_LIB; else export SEHIP_LIB=$PWD/semantic-embeddings_amd/sehip/variants/libsehip_$v.so; fi
    timeout 600 python tools/bench_kernels.py pdist --reps 7 2>&1 | grep -v amdgpu.ids >> $log
    timeout 900 python -m pytest tests/test_gpu_retrieval.py -x -q -m gpu -k "pairwise or golden or benchmarked or full_size" 2>&1 | tail -2 >> $log
  done
  ;;
e2e)
  # end-to-end evaluation (what evaluate_retrieval.py's main does per --feat file), both branches, three calls each
  unset SEHIP_LIB
  timeout 600 python tools/eval_e2e.py 2>&1 | grep -v amdgpu.ids >> $log
  timeout 600 python tools/eval_e2e.py --euclid 2>&1 | grep -v amdgpu.ids >> $log
  timeout 600 python tools/eval_e2e.py --profile --reps 2 2>&1 | grep -v amdgpu.ids | head -40 >> $log
  timeout 1200 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -3 >> $log
  ;;
ranktests)
  # product library: ranking tests + fuzz + kernel microbench
  unset SEHIP_LIB
  timeout 2400 python -m pytest tests/test_gpu_retrieval.py -x -q -m gpu >> $log 2>&1
  timeout 400 python tools/fuzz_rank.py --seconds 120 --seed 61 >> $log 2>&1
  timeout 300 python tools/fuzz_rank.py --seconds 40 --seed 62 --long >> $log 2>&1
  timeout 400 python tools/bench_kernels.py rank 2>&1 | grep -v amdgpu.ids >> $log
  ;;
icache)
  # instruction-cache counters of the ranking kernels: tools/gpu_r6.sh icache name...
  shift
  export TMPDIR=/tmp
  rocprofv3 --list-avail 2>/dev/null | grep "Counter_Name" | grep -i "IFETCH\|WAIT_\|ICACHE\|INST_LEVEL" >> $log
  for v in "$@"; do
    for ctr in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
      OUT=$PWD/gpurun_out/r6ic_$v; rm -rf $OUT; mkdir -p $OUT
      ( cd /tmp && SEHIP_LIB=$GRAFT_REPO_ROOT/semantic-embeddings_amd/sehip/variants/libsehip_$v.so timeout 600 rocprofv3 --pmc $ctr --output-format csv -d $OUT/p -o x -- python $GRAFT_REPO_ROOT/tools/dev_img.py time --reps 2 > $OUT/run.log 2>&1 )
      echo "== $v : $ctr" >> $log
      find $OUT/p -name "*counter_collection.csv" | head -1 | xargs -I{} python tools/pmc_summary.py {} 2>&1 | grep -A3 "rank_rows_reg_kernel<98, false, true, [23]" >> $log
      rm -rf $OUT
    done
  done
  ;;
kt)
  # per-kernel durations of the ranking call (rocprofv3 kernel trace): tools/gpu_r6.sh kt name
  shift
  export TMPDIR=/tmp
  for v in "$@"; do
    OUT=$PWD/gpurun_out/r6kt_$v; rm -rf $OUT; mkdir -p $OUT
    ( cd /tmp && SEHIP_LIB=$GRAFT_REPO_ROOT/semantic-embeddings_amd/sehip/variants/libsehip_$v.so rocprofv3 --kernel-trace --stats -d $OUT/prof -o kt -- python $GRAFT_REPO_ROOT/tools/dev_img.py time --reps 3 > $OUT/run.log 2>&1 )
    DB=$(find $OUT/prof -name "*.db" | head -1)
    echo "== $v" >> $log
    [ -n "$DB" ] && python tools/rocprof_summary.py $DB "dev_img.py time ($v)" 2>&1 | head -30 >> $log
    rm -rf $OUT/prof
  done
  ;;
ab)
  # A/B timing of ranking-kernel variants: tools/gpu_r6.sh ab name1 name2 ...  (libsehip_<name>.so under sehip/variants)
  shift
  for rep in 1 2; do
  for v in "$@"; do
    echo "== $v" >> $log
    SEHIP_LIB=$PWD/semantic-embeddings_amd/sehip/variants/libsehip_$v.so timeout 300 python tools/dev_img.py time --reps 7 2>&1 | grep "^rank" >> $log
  done
  done
  ;;
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
evidence)
  # end-of-round evidence: bench line (both branches), kernel microbenchmarks, rocprofv3 kernel stats of the bench command,
  # phase profile of the image path.  Output: gpurun_out/r6ev/
  unset SEHIP_LIB
  OUT=gpurun_out/r6ev; mkdir -p $OUT; export TMPDIR=/tmp
  ( timeout 1500 python bench.py --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err; head -c 300 $OUT/bench.json >> $log; echo >> $log
  ( timeout 900 python bench.py --steps 20 --warmup 5 --metric euclid --no-train --no-sharded --no-cpu-baseline ) > $OUT/bench_euclid.json 2>> $OUT/bench.err
  for what in pdist rank fused shard hprec rownorm; do timeout 400 python tools/bench_kernels.py $what 2>&1 | grep -v amdgpu.ids; done > $OUT/kernels.log 2>&1
  ( SEHIP_LIB=$PWD/semantic-embeddings_amd/sehip/libsehip_tuning.so SE_RANK_PEEL=3 SE_RR_PROFILE=1 timeout 300 python tools/dev_img.py time --reps 1 ) 2>&1 | grep -v amdgpu.ids > $OUT/rank_phase_profile.txt
  ( SEHIP_LIB=$PWD/semantic-embeddings_amd/sehip/libsehip_tuning.so SE_RANK_PEEL=2 SE_RR_PROFILE=1 timeout 300 python tools/dev_img.py time --reps 1 ) 2>&1 | grep -v amdgpu.ids >> $OUT/rank_phase_profile.txt
  ( timeout 600 python tools/topk_skew.py; timeout 600 python tools/topk_skew.py --n 160146 --q 20000 --d 1000 --classes 125 ) 2>&1 | grep -v amdgpu.ids > $OUT/topk_skew.txt
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r6 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
  cd $GRAFT_REPO_ROOT
  DB=$(find $OUT/prof -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train" > $OUT/prof_summary.txt
  rm -rf $OUT/prof
  cat $OUT/kernels.log >> $log
  ;;
pmcrk)
  # FETCH / WRITE of the ranking kernels only (dev library)
  OUT=gpurun_out/r6pmc2; mkdir -p $OUT; export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/p_$c -o x -- python tools/dev_img.py time --reps 2 > $OUT/p_$c.log 2>&1
    find $OUT/p_$c -name "*counter_collection.csv" | head -1 | xargs -I{} python tools/pmc_summary.py {} 2>&1 | grep -A2 "rank_rows_reg_kernel<98, false, true, [23]" >> $log
    rm -rf $OUT/p_$c
  done
  timeout 300 python tools/dev_img.py time >> $log 2>&1
  SE_RANK_PEEL=3 timeout 300 python tools/dev_img.py check 2>&1 | tail -1 >> $log
  ;;
pmc)
  # PMC counter passes, each in its own rocprofv3 run with no tracing flags (MI355X_MICROARCH.md): the headline kernels
  unset SEHIP_LIB
  OUT=gpurun_out/r6pmc; mkdir -p $OUT; export TMPDIR=/tmp
  run_pmc () { # name counters cmd...
    local name=$1; local ctr=$2; shift 2
    timeout 600 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$name -o $name -- "$@" > $OUT/pmc_$name.log 2>&1
    find $OUT/pmc_$name -name "*counter_collection.csv" | head -1 | xargs -I{} python tools/pmc_summary.py {} > $OUT/pmc_$name.txt 2>&1
    rm -rf $OUT/pmc_$name
  }
  RK="python tools/dev_img.py time --reps 2"
  PD="python tools/bench_kernels.py pdist --reps 2"
  run_pmc rk_fetch "FETCH_SIZE GRBM_GUI_ACTIVE" $RK
  run_pmc rk_write "WRITE_SIZE" $RK
  run_pmc rk_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" $RK
  run_pmc rk_sq2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT" $RK
  run_pmc pd_fetch "FETCH_SIZE GRBM_GUI_ACTIVE" $PD
  run_pmc pd_write "WRITE_SIZE" $PD
  run_pmc pd_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" $PD
  run_pmc pd_sq2 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" $PD
  for f in $OUT/pmc_*.txt; do echo "== $f" >> $log; cat $f >> $log; done
  ;;
alltests)
  unset SEHIP_LIB
  timeout 3400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 >> $log
  ;;
fuzz)
  unset SEHIP_LIB
  timeout 400 python tools/fuzz_rank.py --seconds 150 --seed 21 >> $log 2>&1
  timeout 400 python tools/fuzz_rank.py --seconds 60 --seed 22 --long >> $log 2>&1
  timeout 400 python tools/fuzz_topk.py --seconds 100 >> $log 2>&1
  timeout 400 python tools/fuzz_hprec.py --seconds 40 >> $log 2>&1
  ;;
refprof)
  # phase profile of pf_refine_kernel at the all-pairs size and at the shard size (tuning library)
  export SEHIP_LIB=$PWD/semantic-embeddings_amd/sehip/libsehip_tuning.so
  SE_TOPK_VERBOSE=1 timeout 300 python tools/bench_kernels.py fused --reps 3 2>&1 | grep -v amdgpu.ids | sort | uniq -c >> $log
  SE_TOPK_VERBOSE=1 timeout 300 python tools/bench_kernels.py shard --reps 2 2>&1 | grep -v amdgpu.ids | sort | uniq -c >> $log
  ;;
pdprof)
  export SEHIP_LIB=$PWD/semantic-embeddings_amd/sehip/libsehip_tuning.so
  SE_PD_PROFILE=1 timeout 300 python tools/bench_kernels.py pdist --reps 2 2>&1 | grep -v amdgpu.ids | sort | uniq -c | sort -rn | head -40 >> $log
  for a in 1 2 4; do echo "== SE_PD_ABLATE=$a" >> $log; SE_PD_ABLATE=$a timeout 300 python tools/bench_kernels.py pdist --reps 5 2>&1 | grep -v amdgpu.ids >> $log; done
  ;;
tests)
  unset SEHIP_LIB
  timeout 2400 python -m pytest tests/test_gpu_retrieval.py -x -q -m gpu -k "rank or full_size or benchmarked or golden" >> $log 2>&1
  echo "== fuzz_rank" >> $log
  timeout 400 python tools/fuzz_rank.py --seconds 150 --seed 5 >> $log 2>&1
  echo "== bench" >> $log
  timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r6_bench_line.json 2>> $log
  cat gpurun_out/r6_bench_line.json >> $log
  ;;
esac
tail -60 $log
