#!/bin/bash
set -u
OUT=gpurun_out/r4ah; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 600 python bench.py --steps 5 --warmup 1 --metric euclid --no-train --no-sharded --no-cpu-baseline ) > $OUT/bench_euclid.json 2> $OUT/bench.err; tail -1 $OUT/bench.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r4ah/bench_euclid.json").read().strip().splitlines()[-1])
print(json.dumps({"ms_per_step": r["ms_per_step"], "value": r["value"], "verified": r["verified"], "retrieve_topk": (r.get("retrieve_topk") or {}).get("ms"),
                  "kernels": {k: v["ms"] for k, v in r["kernels"].items()}, "config": r["config"].get("workload")}, indent=1))
PY
