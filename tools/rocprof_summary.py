#!/usr/bin/env python
"""Turn a rocprofv3 rocpd SQLite database (what `rocprofv3 --kernel-trace --stats` writes on this
image) into the plain-text per-kernel summary that is committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db "command line that was profiled" > profiles/rNN_xxx.txt
"""
import sqlite3
import sys


def main():
    db, cmd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    c = sqlite3.connect(db)
    print("# rocprofv3 --kernel-trace --stats summary")
    print("# command : %s" % cmd)
    print("# source  : %s" % db)
    print("# durations in microseconds (rocpd `top_kernels` / `kernels` views)\n")
    print("%-8s %14s %14s %8s  %s" % ("calls", "total_us", "avg_us", "pct", "kernel"))
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-8d %14.1f %14.1f %7.2f%%  %s" % (calls, total, avg, pct, name[:150]))
    print("\n# per-kernel launch geometry / resources (first dispatch of each kernel)")
    print("%-60s %12s %10s %6s %6s %6s %9s %8s" % ("kernel", "grid_x", "wg_x", "vgpr", "agpr", "sgpr", "lds_B", "scratch"))
    seen = set()
    for row in c.execute("select name,grid_x,workgroup_x,vgpr_count,accum_vgpr_count,sgpr_count,lds_size,scratch_size from kernels order by start"):
        if row[0] in seen:
            continue
        seen.add(row[0])
        print("%-60s %12d %10d %6d %6d %6d %9d %8d" % ((row[0][:60],) + tuple(row[1:])))


if __name__ == "__main__":
    main()
