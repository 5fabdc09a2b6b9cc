cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05b
run() { # name env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --config objects8 --steps 20 --warmup 5 --no-solo > gpurun_out/r05b/ab2_$name.json 2> gpurun_out/r05b/ab2_$name.err
  python - <<P
import json
try:
    d=json.loads(open("gpurun_out/r05b/ab2_$name.json").read().strip().splitlines()[-1]); print("$name", d["value"], d["ms_per_step"], d.get("tracked_ok"))
except Exception as e: print("ERR $name", e)
P
}
run off PXT_BATCH_RENDERS=0
run m512_s512 PXT_BATCH_RENDERS=1 PXT_NGP_BATCH_GRID=512
run m2048_s512 PXT_BATCH_RENDERS=1 PXT_NGP_BATCH_GRID=2048 PXT_NGP_BATCH_GRID_SHADE=512
run m512_s1024 PXT_BATCH_RENDERS=1 PXT_NGP_BATCH_GRID=512 PXT_NGP_BATCH_GRID_SHADE=1024
run m1024_s512 PXT_BATCH_RENDERS=1 PXT_NGP_BATCH_GRID=1024 PXT_NGP_BATCH_GRID_SHADE=512
run m768_s768 PXT_BATCH_RENDERS=1 PXT_NGP_BATCH_GRID=768
run m384_s384 PXT_BATCH_RENDERS=1 PXT_NGP_BATCH_GRID=384
run m512_s384 PXT_BATCH_RENDERS=1 PXT_NGP_BATCH_GRID=512 PXT_NGP_BATCH_GRID_SHADE=384
run off2 PXT_BATCH_RENDERS=0
