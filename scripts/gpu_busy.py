"""GPU busy fraction of a rocprofv3 kernel trace (rocpd database): union of the kernels' intervals over the span of the
last N dispatches of a marker kernel.  python scripts/gpu_busy.py results.db [marker=lm_refine] [n=30]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "lm_refine"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = c.execute("select name,start,end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if marker in r[0]]
a, b = idx[-n - 1], idx[-1]
seg = rows[a + 1:b + 1]
t0, t1 = rows[a][2], rows[b][2]
busy, cur_s, cur_e = 0, None, None
for s, e in sorted((r[1], r[2]) for r in seg):
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"{n} frames: {(t1 - t0) / n / 1e3:.1f} us per frame, GPU busy {busy / n / 1e3:.1f} us per frame ({100.0 * busy / (t1 - t0):.0f} %), {len(seg) / n:.0f} dispatches per frame")
