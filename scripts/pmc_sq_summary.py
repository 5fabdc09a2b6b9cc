"""Per-kernel SQ counter summary from a rocprofv3 --pmc run (csv output):
    python scripts/pmc_sq_summary.py DIR > profiles/rNN_pmc_sq.json
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over
waves, SQ_BUSY_CYCLES cycles per shader engine (32 of them), SQ_VALU_MFMA_BUSY_CYCLES cycles summed over
the 1024 SIMDs.  Derived: mfma_busy = MFMA cycles / (SIMDs x busy cycles), valu_busy likewise."""
import collections
import csv
import glob
import json
import sys


def main(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
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
        busy = rec.get("SQ_BUSY_CYCLES", 0.0) / 32.0  # per-SE sums -> cycles the shader array was busy
        if busy > 0:
            simd_cycles = busy * 1024.0
            if "SQ_VALU_MFMA_BUSY_CYCLES" in rec:
                rec["mfma_busy"] = rec["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
            if "SQ_ACTIVE_INST_VALU" in rec:
                rec["valu_busy"] = 4.0 * rec["SQ_ACTIVE_INST_VALU"] / simd_cycles
            if "SQ_WAVE_CYCLES" in rec:
                rec["mean_waves_per_simd"] = 4.0 * rec["SQ_WAVE_CYCLES"] / simd_cycles
        out[k] = rec
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1])
