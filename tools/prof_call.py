"""Ordered listing of the last N dispatches of a rocprofv3 rocpd database (kernel-trace): one line per dispatch with its duration,
the idle gap in front of it and its workgroup count -- what ONE sampler call is made of, launch by launch.
usage: python tools/prof_call.py <db> [N=200]"""
import sqlite3, sys
db = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 200
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, (grid_x/workgroup_x)*(grid_y/workgroup_y)*(grid_z/workgroup_z), workgroup_x*workgroup_y*workgroup_z "
                 "from kernels order by start").fetchall()[-n_last:]
t0 = rows[0][1]
prev = None
print("# t_us (from the first listed dispatch) | dur_us | gap_before_us | workgroups x threads | kernel")
for name, st, en, wg, th in rows:
    gap = 0.0 if prev is None else (st - prev) / 1e3
    print(f"{(st - t0) / 1e3:9.1f} | {(en - st) / 1e3:7.2f} | {gap:6.2f} | {wg:5d} x {th:4d} | {name[:90]}")
    prev = en
