#!/bin/bash
# scratch experiment driver (gpurun)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; out=$R/gpurun_out
rocprofv3 --kernel-trace -d /tmp/kty -o kt -- python $R/scripts/bench_ycb.py 70 > /tmp/y.log 2>&1
db=$(find /tmp/kty -name '*.db' | head -1)
{ head -1 /tmp/y.log; python $R/scripts/gpu_busy.py $db lm_refine 40; python $R/scripts/frame_timeline.py $db 30; } > $out/r03_ycb_frame_timeline.txt 2>&1
python $R/scripts/rocpd_summary.py $db | head -30 > $out/r03_ycb_kernel_stats.csv
rocprofv3 --kernel-trace -d /tmp/ktb -o kt -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > /tmp/b.log 2>&1
db=$(find /tmp/ktb -name '*.db' | head -1)
{ python $R/scripts/gpu_busy.py $db lm_refine 30; python $R/scripts/frame_timeline.py $db 25; } > $out/r03_frame_timeline.txt 2>&1
