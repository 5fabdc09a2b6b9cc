"""Per-kernel HBM-side traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are
collected in separate runs: TCC has 4 counter slots, FETCH_SIZE takes 3, WRITE_SIZE 2 --
MI355X_MICROARCH.md "rocprofv3 PMC slots").  Counters are in KiB.  Caveat from the same guide:
on gfx950 FETCH_SIZE reads exactly half the bytes of a WIDE coalesced (16 B/lane) stream and is
uncalibrated for other access widths; the hash-grid gathers are 4 B/lane scattered reads, so the
raw value is reported (no x2) next to the algorithmic figure.
Usage: python scripts/pmc_summary.py fetch.db write.db > profiles/rNN_pmc.json"""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    c = sqlite3.connect(path)
    q = ("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? "
         "group by kernel_name")
    return {r[0].split("(")[0]: (r[1], r[2]) for r in c.execute(q, (counter,))}


def main(fetch_db, write_db):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith(("pxt::", "void pxt::", "_ZN3pxt")):
            continue
        nf, vf = f.get(k, (0, 0.0))
        nw, vw = w.get(k, (0, 0.0))
        out[k] = {"launches": max(nf, nw), "fetch_bytes_per_launch": vf * 1024, "write_bytes_per_launch": vw * 1024}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
