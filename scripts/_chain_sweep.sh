mkdir -p gpurun_out/r06
cd /root/repo
L=gpurun_out/r06/render_sweep8.log
: > $L
run() { env "$@" python scripts/bench_render_chain.py 4 2>&1 | grep -E "knobs|digests|rror" | cut -c1-330 >> $L; }
run A=1
cat $L
python scripts/variant_checksum.py 320 240 | grep -v unet; python scripts/variant_checksum.py 640 480 | grep -v unet
