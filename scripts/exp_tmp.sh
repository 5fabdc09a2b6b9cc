cd $GRAFT_REPO_ROOT
run() { printf "%-44s" "[$1]"; PXT_CONV_PLAN="$1" python scripts/unet_pass_timeline.py 2>&1 | grep "host ahead" | sed 's/two-image pass, host ahead://'; }
for rep in 1 2 3; do
run ""
run "16:20:0"
run "15:21:0;16:20:0"
run "14:21:0;15:21:0;16:20:0"
done
PXT_CONV_PLAN="14:21:0;15:21:0;16:20:0" python -m pytest tests/test_unet_gpu.py -q -k "not every_tile" 2>&1 | tail -3
