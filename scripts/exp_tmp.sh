#!/bin/bash
# scratch experiment driver (gpurun)
cd /root/repo
for gb in 0 0.008 0.03 0.1 0.3 1.0 16 0; do
  echo "== PXT_NGP_BOX_GB=$gb"
  PXT_NGP_BOX_GB=$gb timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['stage_ms_per_frame']['nerf_render'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['extras']['value_k200']['frames_per_s'])"
done
