# usage: bash scripts/ab_lib.sh name1 name2 ...   (pixtrack_amd/libpxt_<name>.so; "base" = the product library)
for v in "$@"; do
  if [ $v = base ]; then unset PIXTRACK_HIP_LIB; else export PIXTRACK_HIP_LIB=$GRAFT_REPO_ROOT/pixtrack_amd/libpxt_$v.so; fi
  for rep in 1 2; do echo "== $v"; python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_frame'], d['tracked_ok'], d['roofline']['frac'])"; done
done
