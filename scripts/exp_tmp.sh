cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ngp_gpu.py tests/test_variants_gpu.py tests/test_fullsize_golden_gpu.py tests/test_sequence_golden_gpu.py tests/test_ycb_gpu.py -q 2>&1 | tail -5
timeout 300 python scripts/tail_rays.py 2>&1 | grep frame
PXT_NGP_ROUNDS=2 timeout 300 python scripts/tail_rays.py 2>&1 | grep frame | head -4
