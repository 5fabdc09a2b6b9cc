cd $GRAFT_REPO_ROOT
python -m pytest tests/test_variants_gpu.py -q 2>&1 | tail -3
b() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_frame'], d['roofline']['frac'], d['tracked_ok'])"; }
for rep in 1 2 3; do
echo "== default"; b
echo "== xcd bands"; PXT_NGP_XCD_BANDS=1 b
done
