#!/bin/bash
# Round-3 evidence session: full GPU test suite, bench line, kernel microbenchmarks, rocprofv3 kernel stats of the bench command,
# PMC passes (separate runs, no tracing flags) of the fused top-k and the ranking / distance kernels.  Output: gpurun_out/r3ev/
set -u
OUT=gpurun_out/r3ev; mkdir -p $OUT; export TMPDIR=/tmp
nproc > $OUT/host.txt; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/host.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
( timeout 900 python bench.py --steps 5 --warmup 1 ) > $OUT/bench.json 2> $OUT/bench.err; head -c 600 $OUT/bench.json; echo
( timeout 600 python bench.py --steps 5 --warmup 1 --metric euclid --no-train --no-sharded --no-cpu-baseline ) > $OUT/bench_euclid.json 2>> $OUT/bench.err
for what in pdist rank topk fused shard loss hprec; do timeout 400 python tools/bench_kernels.py $what 2>&1 | grep -v amdgpu.ids; done > $OUT/kernels.log 2>&1; cat $OUT/kernels.log
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r3 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train" > $OUT/prof_summary.txt && head -40 $OUT/prof_summary.txt
rm -rf $OUT/prof
run_pmc () { # name counters cmd...
  local name=$1; local ctr=$2; shift 2
  ( cd /tmp; timeout 400 rocprofv3 --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$name -o $name -- "$@" > $GRAFT_REPO_ROOT/$OUT/pmc_$name.log 2>&1 )
  find $OUT/pmc_$name -name "*counter_collection.csv" | head -1 | xargs -I{} python tools/pmc_summary.py {} > $OUT/pmc_$name.txt 2>&1
  echo "== $name: $ctr"; cat $OUT/pmc_$name.txt
}
B="python $GRAFT_REPO_ROOT/tools/bench_kernels.py"
run_pmc fu_fetch "FETCH_SIZE GRBM_GUI_ACTIVE" $B fused --reps 2
run_pmc fu_write "WRITE_SIZE" $B fused --reps 2
run_pmc fu_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" $B fused --reps 2
run_pmc rk_fetch "FETCH_SIZE GRBM_GUI_ACTIVE" $B rank --reps 2
run_pmc rk_write "WRITE_SIZE" $B rank --reps 2
run_pmc rk_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" $B rank --reps 2
run_pmc pd_fetch "FETCH_SIZE GRBM_GUI_ACTIVE" $B pdist --reps 2
run_pmc pd_write "WRITE_SIZE" $B pdist --reps 2
run_pmc pd_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" $B pdist --reps 2
run_pmc sh_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" $B shard --reps 2 --q 16384
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
