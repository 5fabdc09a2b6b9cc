#!/bin/bash
# scratch experiment driver (gpurun)
cd /root/repo
PIXTRACK_HIP_LIB=/root/repo/pixtrack_amd/libpxt_stamps.so python scripts/lm_stamps.py 128 2>&1 | tail -30
python scripts/bench_lm.py 2>&1 | tail -12
