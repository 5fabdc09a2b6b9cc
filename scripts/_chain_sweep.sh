mkdir -p gpurun_out/r06
cd /root/repo
export TMPDIR=/tmp
L=gpurun_out/r06/chain_sweep3.log
: > $L
run() { env "$@" python scripts/bench_render_chain.py 4 2>&1 | grep knobs | cut -c1-330 >> $L; }
export PXT_NGP_ROUNDS=0 PXT_NGP_SKEW=0
run PXT_NGP_PIPES=1 PXT_NGP_TAIL_GRID=2048
run PXT_NGP_PIPES=1 PXT_NGP_TAIL_GRID=3072
run PXT_NGP_PIPES=1 PXT_NGP_TAIL_GRID=4096
run PXT_NGP_PIPES=1 PXT_NGP_TAIL_GRID=6144
run PXT_NGP_PIPES=1 PXT_NGP_TAIL_GRID=8192
run PXT_NGP_PIPES=2 PXT_NGP_TAIL_GRID=2048
run PXT_NGP_PIPES=2 PXT_NGP_TAIL_GRID=4096
run PXT_NGP_PIPES=1 PXT_NGP_TAIL_GRID=4096 PXT_NGP_TAIL_DIV=16
run PXT_NGP_PIPES=1 PXT_NGP_TAIL_GRID=4096 PXT_NGP_TAIL_DIV=256
run PXT_NGP_PIPES=1 PXT_NGP_TAIL_GRID=4096 PXT_NGP_G_COMPACT=2048
run PXT_NGP_PIPES=1 PXT_NGP_TAIL_GRID=4096 PXT_NGP_G_COMPACT=512
cat $L
cd /tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r06/kt_r0 -o kt -- env PXT_NGP_PIPES=1 PXT_NGP_TAIL_GRID=4096 python /root/repo/scripts/bench_render_chain.py 1 > /dev/null 2>&1
