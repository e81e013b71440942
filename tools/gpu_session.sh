#!/bin/bash
# One GPU-box evidence session: GPU tests, bench line, kernel microbenchmarks, rocprof kernel summary, PMC traffic of the
# ranking kernel.  Output under gpurun_out/$1/   (set ABLATE=1 to add the pdist ablation / store-pattern probes)
set -u
TAG=${1:-s1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/host.txt; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/host.txt
python -c "import numpy; numpy.show_config()" 2>&1 | grep -iE "name|openblas|version" >> $OUT/host.txt
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
( timeout 900 python bench.py --steps 5 --warmup 1 ) > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
for what in pdist rank topk loss hprec; do timeout 300 python tools/bench_kernels.py $what; done > $OUT/kernels.log 2>&1
cat $OUT/kernels.log
TUNING=$PWD/semantic-embeddings_amd/sehip/libsehip_tuning.so     # the product library ignores every tuning / ablation switch
if [ "${ABLATE:-0}" = "1" ]; then
  for ab in 1 2 3; do echo "SE_PD_ABLATE=$ab"; SEHIP_LIB=$TUNING SE_PD_ABLATE=$ab timeout 300 python tools/bench_kernels.py pdist; done > $OUT/ablate.log 2>&1
  SEHIP_LIB=$TUNING SE_PD_NOSTAGGER=1 timeout 300 python tools/bench_kernels.py pdist >> $OUT/ablate.log 2>&1
  cat $OUT/ablate.log
  hipcc --offload-arch=gfx950 -O3 tools/probes/store_patterns.hip -o /tmp/store_patterns && timeout 120 /tmp/store_patterns > $OUT/store_patterns.log 2>&1
  cat $OUT/store_patterns.log
fi
SEHIP_LIB=$TUNING SE_RR_PROFILE=1 timeout 200 python tools/bench_kernels.py rank --reps 2 2>&1 | grep profile | tail -2 > $OUT/rank_phase_profile.txt; cat $OUT/rank_phase_profile.txt
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train > $OUT/prof.log 2>&1
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train" > $OUT/prof_summary.txt && cat $OUT/prof_summary.txt
find $OUT/prof -name "*.db" -size +20M -delete
# HBM traffic of the ranking kernel: separate --pmc passes, no tracing flags
run_pmc () { # name counters cmd...
  local name=$1; local ctr=$2; shift 2
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$name -o $name -- "$@" > $OUT/pmc_$name.log 2>&1
  find $OUT/pmc_$name -name "*counter_collection.csv" | head -1 | xargs -I{} python tools/pmc_summary.py {} > $OUT/pmc_$name.txt 2>&1
  cat $OUT/pmc_$name.txt
}
run_pmc rk_fetch "FETCH_SIZE GRBM_GUI_ACTIVE" python tools/bench_kernels.py rank --reps 2
run_pmc rk_write "WRITE_SIZE" python tools/bench_kernels.py rank --reps 2
run_pmc rk_sq2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" python tools/bench_kernels.py rank --reps 2
find $OUT -name "*.csv" -size +5M -delete
