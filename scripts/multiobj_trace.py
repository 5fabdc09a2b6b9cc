"""Per-step summary of a rocprofv3 kernel trace (rocpd database) of a lock-step multi-object run
(scripts/bench_multiobj.py / bench.py --config objects8): wall time per step, GPU busy (union of all queues), summed kernel
time per family, the largest kernels.   python scripts/multiobj_trace.py results.db [steps=8] [launches_per_step=1] [skip_last=0]
(skip_last: trailing lm_refine_batch launches to leave out - bench.py --config objects8 ends with 4 one-group diagnostic steps
with render-ahead off)"""
import sqlite3
import sys
from collections import defaultdict


def fam(n):
    for key, name in (("lm_refine_batch", "lm_batch"), ("lm_refine", "lm"), ("ngp_render_kernel", "ngp_render"), ("ngp_raygen", "ngp_raygen"),
                      ("ngp_", "ngp_other"), ("conv3x3", "conv"), ("conv_first", "conv"), ("head_", "heads"),
                      ("splitk", "conv_splitk"), ("maxpool", "conv_pool"), ("sample_sparse", "sample"), ("depth_mask", "mask")):
        if key in n:
            return name
    return "other"


def main(path, steps=8, per_step=1, skip_last=0):
    c = sqlite3.connect(path)
    rows = c.execute("select name,start,end,queue_id from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if "lm_refine_batch" in r[0]]
    if skip_last:
        idx = idx[:-skip_last]
    n_l = steps * per_step
    a, b = idx[-n_l - 1], idx[-1]
    seg = rows[a + 1:b + 1]
    t0, t1 = rows[a][2], rows[b][2]
    busy, cur_s, cur_e = 0, None, None
    for s, e in sorted((r[1], r[2]) for r in seg):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"{steps} steps: {(t1 - t0) / steps / 1e6:.3f} ms per step, GPU busy {busy / steps / 1e6:.3f} ms per step "
          f"({100.0 * busy / (t1 - t0):.0f} %), {len(seg) / steps:.0f} dispatches per step, queues {sorted({r[3] for r in seg})}")
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in seg:
        tot[fam(r[0])] += r[2] - r[1]
        cnt[fam(r[0])] += 1
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print(f"  {k:12s} {v / steps / 1e6:8.3f} ms per step  {cnt[k] / steps:7.1f} launches")
    ker = defaultdict(float)
    for r in seg:
        ker[r[0].split("(")[0][-70:]] += r[2] - r[1]
    for k, v in sorted(ker.items(), key=lambda kv: -kv[1])[:12]:
        print(f"    {v / steps / 1e6:8.3f} ms  {k}")


if __name__ == "__main__":
    main(sys.argv[1], *(int(x) for x in sys.argv[2:]))
