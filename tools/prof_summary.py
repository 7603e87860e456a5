"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel table (markdown/CSV-ish text)."""
import sqlite3, sys
db = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
                 "(grid_x/workgroup_x)*(grid_y/workgroup_y)*(grid_z/workgroup_z), lds_size, vgpr_count, accum_vgpr_count, scratch_size "
                 "from kernels group by name, grid_x, grid_y, grid_z order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
span = c.execute("select (max(end)-min(start))/1e3 from kernels").fetchone()[0]
print(f"# total kernel time {tot:.1f} us over {sum(r[1] for r in rows)} dispatches; first-to-last span {span:.1f} us")
print("name | calls | total_us | pct | avg_us | min_us | max_us | workgroups (grid x*y*z) | lds | vgpr | agpr | scratch")
for r in rows[:40]:
    print(f"{r[0][:70]} | {r[1]} | {r[2]:.1f} | {100*r[2]/tot:.1f} | {r[3]:.2f} | {r[4]:.2f} | {r[5]:.2f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]}")
