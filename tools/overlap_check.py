"""From a rocprofv3 rocpd database: how much do kernels of different streams overlap in time?  usage: overlap_check.py <db>"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = c.execute(f"select start, end, name, {qcol or '0'} from kernels order by start").fetchall()
rows = [r for r in rows if r[2].startswith("void k_gemm") or r[2].startswith("void k_attn") or r[2].startswith("void k_xattn") or r[2].startswith("void k_head")]
tot = sum(r[1] - r[0] for r in rows)
span = rows[-1][1] - rows[0][0]
ov = 0
last_end = 0
for s, e, n, q in rows:
    if s < last_end:
        ov += min(e, last_end) - s
    last_end = max(last_end, e)
print(f"{len(rows)} kernels on queues {sorted(set(r[3] for r in rows))}: summed duration {tot / 1e6:.2f} ms, span {span / 1e6:.2f} ms, overlapped time {ov / 1e6:.2f} ms ({100 * ov / tot:.1f} % of the summed duration)")
