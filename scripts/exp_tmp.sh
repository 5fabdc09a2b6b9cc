cd $GRAFT_REPO_ROOT
python -m pytest tests/test_unet_gpu.py -q 2>&1 | tail -3
python scripts/bench_conv.py --all-cfgs 2>&1 | grep -E "^(640x480|320x240x2   64)" | grep -E "cfg (2|4|1) "
PIXTRACK_HIP_LIB=$GRAFT_REPO_ROOT/pixtrack_amd/libpxt_stamps.so python scripts/conv_stamps.py 480 640 64 64 2 2>&1 | grep -v amdgpu.ids | head -12
run() { printf "%-22s" "$1"; PIXTRACK_HIP_LIB=$GRAFT_REPO_ROOT/pixtrack_amd/$1 python scripts/unet_pass_timeline.py 2>&1 | grep "host ahead" | sed 's/two-image pass, host ahead://'; }
for rep in 1 2 3; do run libpixtrack_hip.so; done
