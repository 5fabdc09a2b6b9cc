cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_unet_gpu.py tests/test_variants_gpu.py -q -x 2>&1 | tail -5
run() { printf "%-22s" "inlaunch=$1"; PXT_CONV_INLAUNCH_REDUCE=$1 timeout 120 python scripts/unet_pass_timeline.py 2>&1 | grep "host ahead" | sed 's/two-image pass, host ahead://'; }
for rep in 1 2 3; do run 1; run 0; done
