set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05b
timeout 900 python -m pytest tests/test_render_batch_gpu.py -x -q 2>&1 | tail -15
timeout 600 python scripts/bench_render_batch.py > gpurun_out/r05b/render_batch.log 2>&1; tail -8 gpurun_out/r05b/render_batch.log

for i in 1 2; do
for b in 0 1; do PXT_BATCH_RENDERS=$b timeout 600 python bench.py --config objects8 --steps 20 --warmup 5 --no-solo > gpurun_out/r05b/obj8_b${b}_$i.json 2> gpurun_out/r05b/obj8_b${b}_$i.err; python - <<P
import json
try:
    d=json.loads(open("gpurun_out/r05b/obj8_b${b}_$i.json").read().strip().splitlines()[-1]); print("batch=$b", d["value"], d["ms_per_step"], d.get("tracked_ok"))
except Exception as e: print("ERR", e)
P
done; done
