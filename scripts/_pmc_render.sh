# SQ / TA counter passes of the render kernel alone (scripts/bench_render_chain.py 1), cooperative corner fetch on and off
root=/root/repo; out=$root/gpurun_out/r06; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for c in 1 0; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY \
    --kernel-trace --output-format csv -d $out/psq_c$c -- env PXT_NGP_COOP=$c python $root/scripts/bench_render_chain.py 1 > $out/psq_c$c.log 2>&1
  python $root/scripts/pmc_sq_summary.py $out/psq_c$c > $out/r06_render_pmc_sq_coop$c.json
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d $out/pin_c$c -- env PXT_NGP_COOP=$c python $root/scripts/bench_render_chain.py 1 > $out/pin_c$c.log 2>&1
  python $root/scripts/pmc_sq_summary.py $out/pin_c$c > $out/r06_render_pmc_insts_coop$c.json
  rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUFFER_LOAD_WAVEFRONTS_sum GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $out/pta_c$c -- env PXT_NGP_COOP=$c python $root/scripts/bench_render_chain.py 1 > $out/pta_c$c.log 2>&1
  python $root/scripts/pmc_sq_summary.py $out/pta_c$c > $out/r06_render_pmc_ta_coop$c.json
done
rm -rf $out/psq_c* $out/pin_c* $out/pta_c*
tail -3 $out/pta_c1.log
