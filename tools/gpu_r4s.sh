#!/bin/bash
# Round-4 session S: 256 x 256 LDS-DMA filter kernel + staged long-row gather in the refinement -- whole GPU suite, bench line, kernel trace of the shard.
set -u
OUT=gpurun_out/r4s; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
for what in shard fused; do timeout 600 python tools/bench_kernels.py $what --reps 4 2>&1 | grep -v amdgpu.ids; done | tee $OUT/kernels.log
( time timeout 1200 python bench.py --steps 5 --warmup 1 ) > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r4s/bench.json").read().strip().splitlines()[-1])
    keep = {k: r.get(k) for k in ("value", "ms_per_step", "verified")}
    keep["sharded_gallery"] = {k: r["sharded_gallery"].get(k) for k in ("ms", "verified", "error")}
    keep["retrieve_topk"] = (r.get("retrieve_topk") or {}).get("ms")
    keep["hprec"] = (r.get("hierarchical_precision") or {}).get("ms")
    keep["kernels"] = {k: v["ms"] for k, v in r["kernels"].items()}
    print(json.dumps(keep, indent=1))
except Exception as e:
    print("bench parse failed", e)
PY
T=semantic-embeddings_amd/sehip/libsehip_tuning.so
SEHIP_LIB=$T SE_PF_PROFILE=1 SE_TOPK_VERBOSE=1 timeout 300 python tools/bench_kernels.py shard --reps 1 2>&1 | grep -E "pf_big_kernel profile|pf_refine_kernel profile|prefilter:" | sort | uniq | head -6 | tee $OUT/profile.log
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r4s -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py shard --reps 3 > $GRAFT_REPO_ROOT/$OUT/prof_shard.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/prof -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py $DB "bench_kernels.py shard --reps 3" > $OUT/prof_summary.txt && sed -n 6,12p $OUT/prof_summary.txt | cut -c1-130; rm -rf $OUT/prof
