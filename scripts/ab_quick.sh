# usage: bash scripts/ab_quick.sh name1 name2 ...  - render digests, then alternating 40-frame bench runs
for v in "$@"; do
  if [ $v = base ]; then unset PIXTRACK_HIP_LIB; else export PIXTRACK_HIP_LIB=$GRAFT_REPO_ROOT/pixtrack_amd/libpxt_$v.so; fi
  echo "$v: $(python scripts/variant_checksum.py 2>/dev/null | grep -E 'rgba|depth' | tr '\n' ' ')"
done
for rep in 1 2; do for v in "$@"; do
  if [ $v = base ]; then unset PIXTRACK_HIP_LIB; else export PIXTRACK_HIP_LIB=$GRAFT_REPO_ROOT/pixtrack_amd/libpxt_$v.so; fi
  echo "== $v $(python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_frame']['nerf_render'], d['tracked_ok'], d['roofline']['isolated']['avg_launch_ms'])")"
done; done
