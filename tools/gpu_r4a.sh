#!/bin/bash
# Round-4 session A: GPU test suite (new: se_rank_rows_init + graph capture, wave-per-row norms, D=1000 K-block top-k through the
# product library, packed merge), the bf16-graph NaN probe with aligned vs packed parameter slices, the bench line, row-norm microbench.
set -u
OUT=gpurun_out/r4a; mkdir -p $OUT; export TMPDIR=/tmp
nproc > $OUT/host.txt; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/host.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log
timeout 300 python tools/bench_kernels.py rownorm 2>&1 | grep -v amdgpu.ids > $OUT/rownorm.log; cat $OUT/rownorm.log
( timeout 600 python tools/graph_nan_probe.py ) > $OUT/nan_probe.log 2>&1; grep -v "^\[engine\]" $OUT/nan_probe.log | tail -45
( time timeout 1200 python bench.py --steps 5 --warmup 1 ) > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r4a/bench.json").read().strip().splitlines()[-1])
    keep = {k: r.get(k) for k in ("value", "ms_per_step", "verified", "rccl")}
    keep["sharded_gallery"] = {k: r["sharded_gallery"].get(k) for k in ("ms", "verified", "verify_detail", "error")}
    keep["cpu_parity"] = r.get("cpu_baseline", {}).get("same_node_parity")
    keep["cpu_value"] = r.get("cpu_baseline", {}).get("value")
    for leg in ("train", "train_bf16", "train_r50", "train_r50_b128", "train_r50_ilsvrc", "train_r50_ilsvrc_b128"):
        keep[leg] = (r.get(leg) or {}).get("value", (r.get(leg) or {}).get("error"))
    keep["retrieve_topk"] = (r.get("retrieve_topk") or {}).get("ms")
    keep["hprec"] = (r.get("hierarchical_precision") or {}).get("ms")
    keep["kernels"] = {k: v["ms"] for k, v in r["kernels"].items()}
    print(json.dumps(keep, indent=1))
except Exception as e:
    print("bench parse failed", e)
PY
