# usage: bash scripts/ab_conv.sh name1 name2 ...  : bench_conv + bench_unet per library variant ("base" = product library)
for v in "$@"; do
  if [ $v = base ]; then unset PIXTRACK_HIP_LIB; else export PIXTRACK_HIP_LIB=$GRAFT_REPO_ROOT/pixtrack_amd/libpxt_$v.so; fi
  echo "== $v"; python scripts/bench_conv.py 2>/dev/null | sed 's/.*pool 0: *//; s/ us .*//' | tr '\n' ' '; echo
  python scripts/bench_unet.py 2>&1 | grep "640x480"
done
