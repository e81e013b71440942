#!/bin/bash
set -u
OUT=gpurun_out/s4; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -x -k "pairwise or golden or full_size" 2>&1 | tail -3
for ab in 0 4 1; do echo "ABLATE=$ab"; SE_PD_ABLATE=$ab timeout 300 python tools/bench_kernels.py pdist 2>&1 | grep "pdist sym"; done | tee $OUT/pdist_ablate4.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_f -o f -- python tools/bench_kernels.py pdist --reps 1 > /dev/null 2>&1; find $OUT/pmc_f -name "*counter_collection.csv" | head -1 | xargs -I{} python tools/pmc_summary.py {} | grep -A1 "pdist_kernel" | head -4
