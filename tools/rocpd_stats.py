#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd .db (kernel-trace) into a per-kernel stats table (like --stats CSV).
usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/rNN_x_kernel_stats.txt"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                  "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(grid_y) "
                  "from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# rocprofv3 --kernel-trace summary of {sys.argv[1]}  (durations in microseconds)")
print(f"{'kernel':70s} {'calls':>7s} {'total_us':>12s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scr':>5s} {'grid':>12s}")
for n, c, s, a, mn, mx, vg, ag, sg, lds, scr, gx, gy in rows:
    short = re.sub(r"\(.*$", "", n)
    short = re.sub(r"^void ", "", short)[:70]
    print(f"{short:70s} {c:7d} {s/1e3:12.1f} {a/1e3:9.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f} {vg:5d} {ag:5d} {sg:5d} {lds:7d} {scr:5d} {str(gx)+'x'+str(gy):>12s}")
print(f"# total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
