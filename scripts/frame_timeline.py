"""One steady-state frame of a bench kernel trace (rocprofv3 rocpd database), every dispatch with the idle gap
before it: python scripts/frame_timeline.py results.db [frame_index]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,queue_id from kernels order by start").fetchall()
lm = [i for i, r in enumerate(rows) if "lm_refine" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(lm) // 2
a, b = lm[k], lm[k + 1]
t0 = rows[a][2]
busy_end = t0
qs = sorted({r[3] for r in rows[a:b + 1]})
for r in rows[a + 1:b + 1]:
    nm = r[0].split("(")[0].replace("void ", "").replace("pxt::", "")[:46]
    gap = (r[1] - busy_end) / 1e3
    print("%8.1f %7.1f  q%d  %-46s %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, qs.index(r[3]), nm, ("<- idle %.1f us" % gap) if gap > 1.0 else ""))
    busy_end = max(busy_end, r[2])
print("frame %.1f us (LM end to LM end)" % ((rows[b][2] - t0) / 1e3))
