run() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_frame']['nerf_render'], d['tracked_ok'], d['roofline']['isolated']['avg_launch_ms'])"; }
echo "base: $(python scripts/variant_checksum.py 2>/dev/null | grep -E 'rgba|depth|mask' | tr '\n' ' ')"
echo "fr:   $(PXT_NGP_FUSE_ROUND=1 timeout 120 python scripts/variant_checksum.py 2>/dev/null | grep -E 'rgba|depth|mask' | tr '\n' ' ')"
for rep in 1 2; do
echo "== base $(run)"
echo "== fuse_round $(PXT_NGP_FUSE_ROUND=1 run)"
echo "== fuse_round pipes1 $(PXT_NGP_FUSE_ROUND=1 PXT_NGP_PIPES=1 run)"
echo "== fuse_round grid1024 $(PXT_NGP_FUSE_ROUND=1 PXT_NGP_ROUND_GRID=1024 run)"
done
