"""Per-dispatch timeline of the last NeRF render in a rocprofv3 rocpd database (kernel trace):
    python scripts/ngp_timeline.py results.db"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "ngp_resolve" in r[0]]
a = idx[-2] + 1; b = idx[-1]
rows = rows[a:b + 1]
first = next(i for i, r in enumerate(rows) if "ngp_raygen_kernel" in r[0])
rows = rows[first:]
t0 = rows[0][1]
qs = sorted({r[3] for r in rows})
tot = {}
for r in rows:
    nm = r[0].split("(")[0].replace("void ", "").replace("pxt::", "")
    print("%8.1f %7.1f  q%d  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, qs.index(r[3]), nm))
    tot[nm] = tot.get(nm, 0) + (r[2] - r[1]) / 1e3
print("span %.1f us" % ((rows[-1][2] - t0) / 1e3))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("  %-40s %8.1f us" % (k, v))
