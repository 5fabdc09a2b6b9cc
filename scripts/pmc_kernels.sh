# usage: bash scripts/pmc_kernels.sh OUT "cmd..."  : L1/L2 request counters per kernel (separate --pmc passes; no TA_* counters:
# a TA_* pass aborted rocprofv3 on this image)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCC_WRITE_REQ_sum TCC_TAG_STALL_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
           "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- "$@" > $out/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:48]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    if not any(t in k for t in ("ngp", "conv", "lm_", "pxt")): continue
    print(k, {c: round(v[0] / v[1], 1) for c, v in sorted(acc[k].items())}, "n=", max(v[1] for v in acc[k].values()))
PY
