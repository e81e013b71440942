#!/bin/bash
set -u
OUT=gpurun_out/r4e; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o dv -- python $GRAFT_REPO_ROOT/tools/bench_devise.py > $GRAFT_REPO_ROOT/$OUT/prof_devise.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/prof -name "*.db" | head -1); python - "$DB" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
rows = cur.execute("select name, grid_x, (end - start) from kernels order by start").fetchall()
import collections
agg = collections.OrderedDict()
for n, g, d in rows:
    k = (n.split("(")[0][-40:], g)
    agg.setdefault(k, []).append(d)
for k, ds in agg.items():
    ds = sorted(ds)
    print("%-44s grid %8d  calls %4d  median %8.1f us" % (k[0], k[1], len(ds), ds[len(ds)//2] / 1e3))
PY
rm -rf $OUT/prof
