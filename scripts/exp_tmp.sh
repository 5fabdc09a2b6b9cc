#!/bin/bash
# scratch experiment driver (gpurun)
cd /root/repo
timeout 900 python -m pytest tests/test_ycb_gpu.py tests/test_render_ahead_gpu.py -q -x 2>&1 | tail -3
for rep in 1 2; do for ra in 1 0; do echo "== PXT_RENDER_AHEAD=$ra"; PXT_RENDER_AHEAD=$ra python scripts/bench_ycb.py 70 2>&1 | tail -1; done; done
