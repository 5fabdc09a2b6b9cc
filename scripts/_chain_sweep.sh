mkdir -p gpurun_out/r06
cd /root/repo
export TMPDIR=/tmp
L=gpurun_out/r06/render_sweep4.log
: > $L
run() { env "$@" python scripts/bench_render_chain.py 4 2>&1 | grep -E "knobs|digests|rror" | cut -c1-330 >> $L; }
run A=1
run PXT_NGP_GRID=3072
run PXT_NGP_GRID=5120
run PXT_NGP_GRID=4096 PXT_NGP_GRID_DIV=32
run PXT_NGP_PIPES=2 PXT_NGP_GRID=2048
cat $L
python scripts/variant_checksum.py 320 240 | grep -v unet; python scripts/variant_checksum.py 640 480 | grep -v unet
timeout 1200 python -m pytest tests/test_render_batch_gpu.py tests/test_ngp_gpu.py tests/test_render_ahead_gpu.py tests/test_fullsize_golden_gpu.py tests/test_variants_gpu.py tests/test_sequence_golden_gpu.py -x -q -m gpu 2>&1 | tail -15
