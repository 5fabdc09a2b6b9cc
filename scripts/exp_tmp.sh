#!/bin/bash
# scratch experiment driver (gpurun)
cd /root/repo
timeout 900 python -m pytest tests/test_variants_gpu.py tests/test_render_ahead_gpu.py tests/test_ycb_gpu.py tests/test_ngp_gpu.py -q -x 2>&1 | tail -2
for rep in 1 2 3; do python scripts/bench_ycb.py 70 | head -1; done
for rep in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], {k:v['frames_per_s'] for k,v in d['extras'].items()})"; done
