#!/bin/bash
# Round-4 session V: Euclidean problems on the 256 x 256 kernel (tests + microbench) and counter passes of the D = 1000 shard
# (matrix-core busy cycles, LDS conflicts, HBM bytes): separate --pmc passes, no tracing flags.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4v; mkdir -p $OUT
( time timeout 1200 python -m pytest tests/test_gpu_topk.py tests/test_gpu_dropin.py -x -q -k "topk or prefilter or sharded or fused" ) > $OUT/pytest_topk.log 2>&1; grep -v amdgpu.ids $OUT/pytest_topk.log | tail -4
timeout 600 python tools/bench_kernels.py fused --reps 5 2>&1 | grep "fused retrieve" | tee $OUT/kernels.log
cat > /tmp/shard_one.py <<'PY'
import sys; sys.path[:0]=[sys.argv[1]+"/semantic-embeddings_amd", sys.argv[1]]
import numpy as np, torch, sehip
rng = np.random.default_rng(0)
g = torch.from_numpy(rng.standard_normal((160146, 1000)).astype(np.float32)).cuda(); sehip.normalize_rows_(g)
q = torch.from_numpy(rng.standard_normal((50000, 1000)).astype(np.float32)).cuda(); sehip.normalize_rows_(q)
for _ in range(2): sehip.retrieve_topk(q, g, 251, kblocks=[448, 276, 276])
torch.cuda.synchronize()
PY
run_pmc () { local name=$1; local ctr=$2
  ( cd /tmp; timeout 400 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$name -o $name -- python /tmp/shard_one.py $R > $OUT/pmc_$name.log 2>&1 )
  find $OUT/pmc_$name -name "*counter_collection.csv" | head -1 | xargs -I{} python $R/tools/pmc_summary.py {} > $OUT/pmc_$name.txt 2>&1
  echo "== $name: $ctr"; grep -E "pf_big|pf_refine|kernel|Kernel" $OUT/pmc_$name.txt | head -8; }
run_pmc sq "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run_pmc fetch "FETCH_SIZE GRBM_GUI_ACTIVE"
run_pmc write "WRITE_SIZE"
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete; rm -rf $OUT/pmc_*/
