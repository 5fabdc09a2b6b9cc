"""Per-dispatch timeline of one two-image UNet pass from a rocprofv3 rocpd database (kernel trace), both streams:
    python scripts/unet_timeline.py results.db [pass_index]
Columns: start (us from the pass's first kernel), duration, queue, grid, kernel."""
import sqlite3
import sys


def main(path, which=None):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, queue_id from kernels order by start"))
    # a pass opens with the first layer: its own kernel, or (round 3) the 64 -> 64 layer's FIRST variant
    def opens(n):
        return "conv_first" in n or "false, 3, true>" in n
    firsts = [i for i, r in enumerate(rows) if opens(r[0]) and (i == 0 or not opens(rows[i - 1][0]) or rows[i][1] - rows[i - 1][1] > 200000)]
    i0 = firsts[which if which is not None else (2 * len(firsts)) // 3]
    nxt = [i for i in firsts if i > i0]
    i1 = nxt[0] if nxt else len(rows)
    t0 = rows[i0][1]
    total, busy_to = 0.0, 0.0
    for r in rows[i0:i1]:
        if not any(k in r[0] for k in ("conv", "head", "maxpool", "splitk")):
            continue
        nm = r[0].split("(")[0][-46:]
        d = (r[2] - r[1]) / 1e3
        total += d
        busy_to = max(busy_to, (r[2] - t0) / 1e3)
        print(f"{(r[1]-t0)/1e3:9.1f} {d:7.1f} us  q{r[7]}  wgs=({r[3]//max(r[6],1)},{r[4]},{r[5]})  {nm}")
    print(f"sum of kernel durations {total:.1f} us; span {busy_to:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
