#!/usr/bin/env python
"""Per-kernel totals of the LAST `ms` milliseconds of kernel activity in a rocprofv3 rocpd database: the timed steps of a training
leg without the MIOpen solver search / warm-up in front of them (whose naive reference kernels otherwise dominate the statistics).
    python tools/rocprof_window.py results.db 150 "label" > profiles/..."""
import collections
import sqlite3
import sys

db, ms, label = sys.argv[1], float(sys.argv[2]), (sys.argv[3] if len(sys.argv) > 3 else "")
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
t_end = max(r[2] for r in rows)
t0 = t_end - ms * 1e6
sel = [(n, e - s) for n, s, e in rows if s >= t0]
tot = collections.Counter()
cnt = collections.Counter()
for n, d in sel:
    tot[n] += d
    cnt[n] += 1
busy = sum(tot.values())
print("# %s: kernels started in the last %.0f ms of device activity (%d launches, %.1f ms of kernel time)" % (label, ms, len(sel), busy / 1e6))
print("%-7s %12s %8s  %s" % ("calls", "total_us", "pct", "kernel"))
for n, d in tot.most_common(22):
    print("%-7d %12.1f %7.2f%%  %s" % (cnt[n], d / 1e3, 100.0 * d / busy, n[:130]))
