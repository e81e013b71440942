#!/bin/bash
# Round-4 session K: after wg_barrier() (explicit LDS drain before every work-group barrier): the two-stream stress that showed
# unsorted se_topk_rows rows, the two-process sharded test in a loop, then the full GPU suite and the bench line.
set -u
OUT=gpurun_out/r4k; mkdir -p $OUT; export TMPDIR=/tmp
for m in pdist topk; do timeout 600 python tools/stress_one_proc.py 300 $m 2>&1 | grep -E "one process"; done | tee $OUT/stress.log
timeout 600 python tools/stress_one_proc.py 300 topk 600 2>&1 | grep -E "one process" | tee -a $OUT/stress.log
fails=0; for i in $(seq 1 12); do timeout 300 python -m pytest tests/test_gpu_dropin.py -q -k two_processes_real_kernels > $OUT/two_$i.log 2>&1 || { fails=$((fails+1)); tail -5 $OUT/two_$i.log; }; done; echo "two-process test: $fails failures of 12" | tee -a $OUT/stress.log
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; tail -6 $OUT/pytest_gpu.log
( time timeout 1200 python bench.py --steps 5 --warmup 1 ) > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r4k/bench.json").read().strip().splitlines()[-1])
    keep = {k: r.get(k) for k in ("value", "ms_per_step", "verified", "roofline")}
    keep["sharded_gallery"] = {k: r["sharded_gallery"].get(k) for k in ("ms", "verified", "error")}
    for leg in ("train", "train_bf16", "train_r50", "train_r50_b128"):
        keep[leg] = (r.get(leg) or {}).get("value", (r.get(leg) or {}).get("error"))
    keep["retrieve_topk"] = (r.get("retrieve_topk") or {}).get("ms")
    keep["hprec"] = (r.get("hierarchical_precision") or {}).get("ms")
    keep["kernels"] = {k: v["ms"] for k, v in r["kernels"].items()}
    print(json.dumps(keep, indent=1))
except Exception as e:
    print("bench parse failed", e)
PY
