#!/bin/bash
# PMC passes over the hierarchical-precision kernel (separate runs, no tracing flags with --pmc)
set -u
OUT=gpurun_out/${1:-hpprof}; mkdir -p $OUT; export TMPDIR=/tmp
run_pmc () { # name counters cmd...
  local name=$1; local ctr=$2; shift 2
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$name -o $name -- "$@" > $OUT/pmc_$name.log 2>&1
  find $OUT/pmc_$name -name "*counter_collection.csv" | head -1 | xargs -I{} python tools/pmc_summary.py {} > $OUT/pmc_$name.txt 2>&1
  grep -A12 "hprec_kernel" $OUT/pmc_$name.txt
}
run_pmc fetch "FETCH_SIZE GRBM_GUI_ACTIVE" python tools/bench_kernels.py hprec --reps 2 --hp-mode whole
run_pmc write "WRITE_SIZE" python tools/bench_kernels.py hprec --reps 2 --hp-mode whole
run_pmc sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" python tools/bench_kernels.py hprec --reps 2 --hp-mode whole
run_pmc sq2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" python tools/bench_kernels.py hprec --reps 2 --hp-mode whole
run_pmc tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" python tools/bench_kernels.py hprec --reps 2 --hp-mode whole
find $OUT -name "*.csv" -size +5M -delete
