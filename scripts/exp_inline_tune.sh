run() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_frame']['nerf_render'], d['tracked_ok'])"; }
for rep in 1 2; do
for p in 1 2 3; do echo "== pipes $p"; PXT_NGP_PIPES=$p run; done
for g in 512 1024 4096; do echo "== shade grid $g"; PXT_NGP_SHADE_GRID=$g run; done
for v in ig1 ig4 ig8; do echo "== $v"; PIXTRACK_HIP_LIB=$GRAFT_REPO_ROOT/pixtrack_amd/libpxt_$v.so run; done
done
