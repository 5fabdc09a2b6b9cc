mkdir -p gpurun_out/r06
cd /root/repo
export TMPDIR=/tmp
export PXT_NGP_ROUNDS=0 PXT_NGP_BATCH_ROUNDS=0 PXT_NGP_TAIL_GRID=4096 PXT_NGP_PIPES=1 PXT_NGP_BATCH_PIPES=1 PXT_NGP_SKEW=0
python bench.py --extra r9_phone 2>/dev/null | cut -c1-400
PXT_NGP_TAIL_GRID=2048 python bench.py --extra r9_phone 2>/dev/null | cut -c1-200
PXT_RENDER_AHEAD=0 python bench.py --extra r9_phone 2>/dev/null | cut -c1-200
cd /tmp
rocprofv3 --kernel-trace -d /root/repo/gpurun_out/r06/kt_r9 -o kt -- python /root/repo/bench.py --extra r9_phone > /dev/null 2>&1
ls -la /root/repo/gpurun_out/r06/kt_r9
