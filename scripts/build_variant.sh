# usage: bash scripts/build_variant.sh NAME FILE(stem: pxt_unet | pxt_ngp | ...) [-Dflags...]   -> pixtrack_amd/libpxt_NAME.so
# (A/B experiments: one source recompiled with extra flags, linked with the product objects; select with PIXTRACK_HIP_LIB)
set -e
name=$1; stem=$2; shift 2
cd "$(dirname "$0")/.."
extra=""; [ "$stem" = pxt_ngp ] && extra="-ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=on -Wno-comment -Wno-unused-result $extra "$@" \
  -c pixtrack_amd/csrc/$stem.hip -o /tmp/${stem}_$name.o
objs=""
for o in pixtrack_amd/csrc/_obj/*.o; do [ "$(basename $o .o)" = "$stem" ] || objs="$objs $o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/${stem}_$name.o -o pixtrack_amd/libpxt_$name.so
echo built pixtrack_amd/libpxt_$name.so
