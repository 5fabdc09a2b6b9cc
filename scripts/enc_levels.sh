# usage (GPU box): bash scripts/enc_levels.sh  -- needs pixtrack_amd/libpxt_enc_<lo>_<hi>.so built with
#   bash scripts/build_variant.sh enc_<lo>_<hi> pxt_ngp -DPXT_EXP_ENC_LEVELS=$((lo*256+hi))   (lo hi: 0 16, 0 4, 4 8, 8 12, 12 16, 0 10)
cd /tmp; export TMPDIR=/tmp
for v in enc_0_16 enc_0_4 enc_4_8 enc_8_12 enc_12_16 enc_0_10; do
  export PIXTRACK_HIP_LIB=$GRAFT_REPO_ROOT/pixtrack_amd/libpxt_$v.so
  rm -rf /tmp/ke; PXT_NGP_PIPES=1 rocprofv3 --kernel-trace --stats -d /tmp/ke -o ke -- python $GRAFT_REPO_ROOT/scripts/bench_ngp.py > /dev/null 2>&1
  echo "$v $(python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py /tmp/ke/ke_results.db | grep ngp_encode | cut -d, -f2-4)"
done
