# usage: bash scripts/ab_stamps.sh "H W Cin Cout cfg" lib1 lib2 ...
shape=$1; shift
for v in "$@"; do
  export PIXTRACK_HIP_LIB=$GRAFT_REPO_ROOT/pixtrack_amd/libpxt_$v.so
  echo "== $v"; python scripts/conv_stamps.py $shape 2>&1 | grep -v amdgpu.ids
done
