#!/bin/bash
set -u
OUT=gpurun_out/r4af; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 1200 python bench.py --steps 5 --warmup 1 ) > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r4af/bench.json").read().strip().splitlines()[-1])
print(json.dumps({"ms_per_step": r["ms_per_step"], "value": r["value"], "verified": r["verified"], "frac": r["roofline"]["frac"],
                  "sharded": r["sharded_gallery"]["ms"], "sharded_verified": r["sharded_gallery"]["verified"],
                  "retrieve_topk": r["retrieve_topk"]["ms"], "hprec": r["hierarchical_precision"]["ms"]}, indent=1))
PY
