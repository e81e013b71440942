#!/bin/bash
set -u
OUT=gpurun_out/r4g; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/prof -o pp -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py pipe --reps 1 > $GRAFT_REPO_ROOT/$OUT/prof_pipe.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/prof -name "*.db" | head -1); python - "$DB" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
seen=set()
for row in cur.execute("select name,grid_x,workgroup_x,vgpr_count,accum_vgpr_count,sgpr_count,lds_size,scratch_size from kernels order by start"):
    if row[0] in seen: continue
    seen.add(row[0])
    if "pdist" in row[0] or "rank_rows_reg" in row[0]:
        print(row[0][:70], row[1:])
PY
rm -rf $OUT/prof
