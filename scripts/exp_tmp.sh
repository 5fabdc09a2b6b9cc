cd $GRAFT_REPO_ROOT
python -m pytest tests/test_unet_gpu.py tests/test_variants_gpu.py -q 2>&1 | tail -4
run() { printf "%-22s" "fuse_first=$1"; PXT_UNET_FUSE_FIRST=$1 python scripts/unet_pass_timeline.py 2>&1 | grep "host ahead" | sed 's/two-image pass, host ahead://'; }
for rep in 1 2 3; do run 1; run 0; done
