cd $GRAFT_REPO_ROOT
run() { printf "%-14s %-40s" "$1" "[$2]"; PIXTRACK_HIP_LIB=$GRAFT_REPO_ROOT/pixtrack_amd/$1 PXT_CONV_PLAN="$2" python scripts/unet_pass_timeline.py 2>&1 | grep "host ahead" | sed 's/two-image pass, host ahead://'; }
for rep in 1 2 3; do
run libpixtrack_hip.so ""
run libpxt_oldinterp.so ""
run libpixtrack_hip.so "13:17:0;14:17:0;15:17:0;16:16:0"
run libpixtrack_hip.so "16:16:0"
done
python -m pytest tests/test_unet_gpu.py -q 2>&1 | tail -3
