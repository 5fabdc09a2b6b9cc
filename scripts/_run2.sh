cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05b
for g in 512 1024; do PXT_NGP_BATCH_GRID=$g PXT_BATCH_RENDERS=1 timeout 600 python bench.py --config objects8 --steps 20 --warmup 5 --no-solo > gpurun_out/r05b/obj8_g$g.json 2> gpurun_out/r05b/obj8_g$g.err; python - <<P
import json
try:
    d=json.loads(open("gpurun_out/r05b/obj8_g$g.json").read().strip().splitlines()[-1]); print("grid=$g", d["value"], d["ms_per_step"], d.get("tracked_ok"))
except Exception as e: print("ERR", e)
P
done
timeout 900 python bench.py --steps 60 --warmup 10 > gpurun_out/r05b/bench_default.json 2> gpurun_out/r05b/bench_default.err
python - <<P
import json
d=json.loads(open("gpurun_out/r05b/bench_default.json").read().strip().splitlines()[-1]); print("headline", d["value"], d["ms_per_step"], d["stage_ms_per_frame"], {k:(v.get("frames_per_s") if isinstance(v,dict) else v) for k,v in d["extras"].items()})
P
