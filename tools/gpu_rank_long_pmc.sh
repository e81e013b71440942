#!/bin/bash
# HBM traffic of the sorted-runs + merge path (rows of 100,000 columns): separate --pmc passes, no tracing flags
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/rlpmc; mkdir -p $OUT
cat > /tmp/rl_one.py <<'PY'
import sys; sys.path[:0]=[sys.argv[1]+"/semantic-embeddings_amd", sys.argv[1]]
import torch, sehip
x=torch.randn(4096,100000,device="cuda")
for _ in range(2): sehip.rank_rows(x)
torch.cuda.synchronize()
PY
run_pmc () { local name=$1; local ctr=$2
  ( cd /tmp; timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$name -o $name -- python /tmp/rl_one.py $R > $OUT/pmc_$name.log 2>&1 )
  find $OUT/pmc_$name -name "*counter_collection.csv" | head -1 | xargs -I{} python $R/tools/pmc_summary.py {} > $OUT/pmc_$name.txt 2>&1
  echo "== $name: $ctr"; cat $OUT/pmc_$name.txt; }
run_pmc fetch "FETCH_SIZE GRBM_GUI_ACTIVE"
run_pmc write "WRITE_SIZE"
run_pmc sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete
