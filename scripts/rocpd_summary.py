"""Per-kernel summary (calls, total/avg/min/max duration) from a rocprofv3 rocpd SQLite
database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes NAME_results.db).
Usage: python scripts/rocpd_summary.py results.db > profiles/rNN_kernel_stats.csv"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    q = (f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
         f"from kernels group by {name_col} order by 3 desc")
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows) or 1
    print("kernel,calls,total_ms,avg_us,min_us,max_us,pct")
    for name, n, tot, avg, mn, mx in rows:
        short = name.split("(")[0][:90].replace(",", ";")
        print(f"{short},{n},{tot/1e6:.3f},{avg/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f},{100*tot/total:.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
