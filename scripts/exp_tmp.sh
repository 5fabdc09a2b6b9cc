cd $GRAFT_REPO_ROOT
python -m pytest tests/test_unet_gpu.py tests/test_fullsize_golden_gpu.py tests/test_sequence_golden_gpu.py -q 2>&1 | tail -3
run() { printf "%-22s" "$1"; PIXTRACK_HIP_LIB=$GRAFT_REPO_ROOT/pixtrack_amd/$1 python scripts/unet_pass_timeline.py 2>&1 | grep "host ahead" | sed 's/two-image pass, host ahead://'; }
for rep in 1 2 3; do
run libpixtrack_hip.so
run libpxt_oldfirst.so
done
