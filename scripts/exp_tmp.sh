#!/bin/bash
# scratch experiment driver (gpurun)
cd /root/repo
for rep in 1 2; do
for cfg in "2 2048" "3 2048" "3 1024" "4 1024" "4 2048"; do
    set -- $cfg
    echo "== pipes=$1 shade_grid=$2"
    PXT_NGP_PIPES=$1 PXT_NGP_SHADE_GRID=$2 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['stage_ms_per_frame']['nerf_render'], d['extras']['value_k200']['frames_per_s'], d['extras']['value_two_renders']['frames_per_s'])"
done
done
