cd $GRAFT_REPO_ROOT
PXT_CONV_DEBUG=1 python scripts/unet_pass_timeline.py 2>&1 | grep "conv layer" | head -16
run() { printf "%-44s" "[$1]"; PXT_CONV_PLAN="$1" python scripts/unet_pass_timeline.py 2>&1 | grep "host ahead" | sed 's/two-image pass, host ahead://'; }
for rep in 1 2 3; do
run ""
run "10:18:2;11:18:2;12:18:2"
run "10:18:4;11:18:4;12:18:4"
run "10:19:4;11:19:4;12:19:4"
done
python -m pytest tests/test_unet_gpu.py tests/test_fullsize_golden_gpu.py -q 2>&1 | tail -3
