#!/bin/bash
# scratch experiment driver (gpurun)
cd /root/repo
for rep in 1 2; do
for sg in 2048 4096 8192 16384; do
    echo -n "shade_grid=$sg: "
    PXT_NGP_SHADE_GRID=$sg timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['stage_ms_per_frame']['nerf_render'], d['roofline']['frac'])"
done
done
