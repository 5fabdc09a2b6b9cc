# usage (on the GPU box, from the repo root): bash scripts/collect_profiles.sh rNN
# Collects everything profiles/ holds for a round, each in its own rocprofv3 run (kernel trace, then one --pmc pass per
# counter group: FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ_* in a third), and writes the summaries under
# gpurun_out/<rNN>/ (copy the ones to be judged into profiles/).
r=${1:-r05}
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
out=$root/gpurun_out/$r; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
cmd="python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- $cmd > $out/${r}_bench_under_rocprof.log 2>&1
python $root/scripts/rocpd_summary.py $out/kt/kt_results.db > $out/${r}_kernel_stats.csv
python $root/scripts/unet_timeline.py $out/kt/kt_results.db > $out/${r}_unet_timeline.txt 2>&1
python $root/scripts/ngp_timeline.py $out/kt/kt_results.db > $out/${r}_ngp_timeline.txt 2>&1
python $root/scripts/frame_timeline.py $out/kt/kt_results.db > $out/${r}_frame_timeline.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pf -o pf -- $cmd > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pw -o pw -- $cmd > $out/pmc_write.log 2>&1
python $root/scripts/pmc_summary.py $out/pf/pf_results.db $out/pw/pw_results.db > $out/${r}_pmc_traffic.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d $out/psq -- $cmd > $out/pmc_sq.log 2>&1
python $root/scripts/pmc_sq_summary.py $out/psq > $out/${r}_pmc_sq.json
# L1 -> L2 requests / latency and L2 hits per kernel (two passes: TCP and TCC counters do not share a pass)
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $out/pl1 -- $cmd > $out/pmc_l1.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $out/pl2 -- $cmd > $out/pmc_l2.log 2>&1
python $root/scripts/pmc_l2_summary.py $out/pl1 $out/pl2 > $out/${r}_pmc_l2.json
# the two-image UNet pass with the host AHEAD of the device (under the profiler the host otherwise falls behind and the
# two streams stop overlapping): one pass's dispatches on both streams
rocprofv3 --kernel-trace -d $out/ut -o ut -- python $root/scripts/unet_pass_timeline.py > $out/${r}_unet_pass.log 2>&1
python $root/scripts/unet_timeline.py $out/ut/ut_results.db -1 > $out/${r}_unet_pass_timeline.txt 2>&1
# ---- the real-asset frame (the reference's own reference-image shapes) as a measured object
for e in ycb_refshape r9_phone; do
  rocprofv3 --kernel-trace --stats -d $out/rs_$e -o rs -- python $root/bench.py --extra $e > $out/${r}_refshape_${e}.json 2> $out/rs_$e.err
  python $root/scripts/rocpd_summary.py $out/rs_$e/rs_results.db > $out/${r}_refshape_${e}_kernel_stats.csv
  python $root/scripts/frame_timeline.py $out/rs_$e/rs_results.db > $out/${r}_refshape_${e}_frame_timeline.txt 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS \
    --kernel-trace --output-format csv -d $out/rsq_$e -- python $root/bench.py --extra $e > $out/rsq_$e.log 2>&1
  python $root/scripts/pmc_sq_summary.py $out/rsq_$e > $out/${r}_refshape_${e}_pmc_sq.json
done
# ---- the eight objects in lock-step on one GPU (configs[3], N = 1): kernel trace + step summary + SQ pass of the batched passes
o8="python $root/bench.py --config objects8 --steps 12 --warmup 4 --no-solo"
rocprofv3 --kernel-trace --stats -d $out/o8 -o o8 -- $o8 > $out/${r}_objects8_under_rocprof.log 2>&1
python $root/scripts/rocpd_summary.py $out/o8/o8_results.db > $out/${r}_objects8_kernel_stats.csv
python $root/scripts/multiobj_trace.py $out/o8/o8_results.db 6 2 4 > $out/${r}_objects8_step_summary.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d $out/o8sq -- $o8 --groups 1 > $out/o8sq.log 2>&1
python $root/scripts/pmc_sq_summary.py $out/o8sq > $out/${r}_objects8_pmc_sq.json
cd $root
python bench.py --config objects8 > $out/${r}_bench_objects8.json 2> $out/${r}_bench_objects8.err
python scripts/bench_lm_batch.py 2341 > $out/${r}_bench_lm_batch.log 2>&1
python scripts/bench_unet_batch.py --sizes 2,4,8,16 > $out/${r}_bench_unet_batch.log 2>&1
python scripts/bench_conv.py --all-cfgs > $out/${r}_conv_cfgs.log 2>&1
python bench.py > $out/${r}_bench.json 2> $out/${r}_bench.err
python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-extras > $out/${r}_bench_k200.json 2>/dev/null
rm -rf $out/pf $out/pw $out/psq $out/pl1 $out/pl2 $out/ut $out/rs_* $out/rsq_* $out/o8 $out/o8sq  # (databases: too large to carry back)
ls -la $out | head -60
