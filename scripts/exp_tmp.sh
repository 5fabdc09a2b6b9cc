#!/bin/bash
# scratch experiment driver (gpurun)
cd /root/repo
timeout 900 python -m pytest tests/test_variants_gpu.py -q -x 2>&1 | tail -2
for rep in 1 2; do
for cfg in "1024 1" "1024 32" "1024 64" "1024 128" "1024 256" "384 100000" "1536 64"; do
    set -- $cfg
    echo "== tail_grid=$1 div=$2"
    PXT_NGP_TAIL_GRID=$1 PXT_NGP_TAIL_DIV=$2 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['stage_ms_per_frame']['nerf_render'], d['extras']['value_k200']['frames_per_s'], d['extras']['value_two_renders']['frames_per_s'])"
done
done
