"""Kernel-trace timeline of the LAST sampler call in a rocprofv3 rocpd database: per kernel name the mean duration and the
mean idle gap in front of it (start minus the previous kernel's end), i.e. where a launch-bound chain spends its time."""
import sqlite3, sys
db = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 260
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x/workgroup_x from kernels order by start").fetchall()
rows = rows[-n_last:]
stat = {}
prev_end = None
for name, st, en, wg in rows:
    key = (name[:60], wg)
    d = stat.setdefault(key, [0, 0.0, 0.0])
    d[0] += 1; d[1] += (en - st) / 1e3
    if prev_end is not None: d[2] += max(0, st - prev_end) / 1e3
    prev_end = en
span = (rows[-1][2] - rows[0][1]) / 1e3
busy = sum(v[1] for v in stat.values()); gap = sum(v[2] for v in stat.values())
print(f"# last {len(rows)} dispatches: span {span:.1f} us, kernels {busy:.1f} us, gaps {gap:.1f} us")
print("name | workgroups | calls | avg_us | avg_gap_before_us")
for k, v in sorted(stat.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]} | {k[1]} | {v[0]} | {v[1]/v[0]:.2f} | {v[2]/v[0]:.2f}")
