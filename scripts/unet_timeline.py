"""Per-dispatch timeline of one UNet pass from a rocprofv3 rocpd database (kernel trace):
    python scripts/unet_timeline.py results.db [pass_index]"""
import sqlite3
import sys


def main(path, which=None):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = list(c.execute(f"select {name_col}, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"))
    firsts = [i for i, r in enumerate(rows) if "conv_first" in r[0]]
    # a pass starts at a conv_first whose predecessor is not a conv_first
    starts = [i for i in firsts if i == 0 or "conv_first" not in rows[i - 1][0]]
    i0 = starts[which if which is not None else len(starts) // 2]
    t0 = rows[i0][1]
    total = 0.0
    j = i0
    while j < len(rows):
        r = rows[j]
        if j > i0 and "conv_first" in r[0] and "conv_first" not in rows[j - 1][0]:
            break
        if any(k in r[0] for k in ("conv", "head", "maxpool", "splitk")):
            nm = r[0].split("(")[0][-48:]
            d = (r[2] - r[1]) / 1e3
            total += d
            print(f"{(r[1]-t0)/1e3:9.1f} {d:7.1f} us  wgs=({r[3]//r[6]},{r[4]},{r[5]})  {nm}")
        j += 1
    print(f"sum of kernel durations {total:.1f} us; span {(rows[j-1][2]-t0)/1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
