"""Per-kernel, per-launch means of the L1 / L2 request counters from rocprofv3 --pmc csv runs:
    python scripts/pmc_l2_summary.py DIR [DIR ...] > profiles/rNN_pmc_l2.json
(TCP_TCC_READ_REQ: 64-B read requests from the L1s to the L2; _LATENCY: summed cycles; TCC_HIT / TCC_MISS: L2 tag results.)"""
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
out = {}
for k, cs in acc.items():
    if not any(t in k for t in ("pxt", "ngp", "conv", "lm_")):
        continue
    rec = {c: v / max(n, 1) for c, (v, n) in cs.items()}
    rec["launches"] = max(n for _, n in cs.values())
    out[k] = rec
print(json.dumps(out, indent=1, sort_keys=True))
