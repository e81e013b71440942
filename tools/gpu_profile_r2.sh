#!/bin/bash
# Round-2 evidence: rocprofv3 kernel summary of the bench command + PMC traffic passes (separate runs, no tracing flags with --pmc)
set -u
OUT=gpurun_out/${1:-prof2}; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train"
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r2 -- $CMD > $OUT/prof.log 2>&1
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB "$CMD" > $OUT/prof_summary.txt && cat $OUT/prof_summary.txt
grep '^{' $OUT/prof.log | cut -c1-300
find $OUT/prof -name "*.db" -size +20M -delete
run_pmc () { # name counters cmd...
  local name=$1; local ctr=$2; shift 2
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$name -o $name -- "$@" > $OUT/pmc_$name.log 2>&1
  find $OUT/pmc_$name -name "*counter_collection.csv" | head -1 | xargs -I{} python tools/pmc_summary.py {} > $OUT/pmc_$name.txt 2>&1
  cat $OUT/pmc_$name.txt
}
run_pmc rk_fetch "FETCH_SIZE GRBM_GUI_ACTIVE" python tools/bench_kernels.py rank --reps 2
run_pmc rk_write "WRITE_SIZE" python tools/bench_kernels.py rank --reps 2
run_pmc rk_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" python tools/bench_kernels.py rank --reps 2
run_pmc rk_sq2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" python tools/bench_kernels.py rank --reps 2
run_pmc pd_fetch "FETCH_SIZE GRBM_GUI_ACTIVE" python tools/bench_kernels.py pdist --reps 2
run_pmc pd_write "WRITE_SIZE" python tools/bench_kernels.py pdist --reps 2
run_pmc pd_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" python tools/bench_kernels.py pdist --reps 2
find $OUT -name "*.csv" -size +5M -delete
