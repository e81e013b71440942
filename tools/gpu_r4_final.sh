#!/bin/bash
# Round-4 final check at HEAD: full GPU suite, bench line, fused / shard microbench.
set -u
OUT=gpurun_out/r4fin; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
( timeout 1200 python bench.py --steps 5 --warmup 1 ) > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
for what in fused shard; do timeout 400 python tools/bench_kernels.py $what 2>&1 | grep -v amdgpu.ids; done | tee $OUT/kernels.log
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r4fin/bench.json").read().strip().splitlines()[-1])
print(json.dumps({"ms_per_step": r["ms_per_step"], "value": r["value"], "verified": r["verified"], "frac": r["roofline"]["frac"],
                  "sharded": r["sharded_gallery"]["ms"], "sharded_verified": r["sharded_gallery"]["verified"],
                  "retrieve_topk": r["retrieve_topk"]["ms"], "hprec": r["hierarchical_precision"]["ms"]}, indent=1))
PY
