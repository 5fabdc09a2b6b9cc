cd $GRAFT_REPO_ROOT
python -m pytest tests/test_unet_gpu.py tests/test_bench_gpu.py -q -x 2>&1 | tail -4
for rep in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_frame'], d['roofline']['frac'], d['tracked_ok'], {k:v['frames_per_s'] for k,v in d['extras'].items()})"; done
