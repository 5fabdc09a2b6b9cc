root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
out=$root/gpurun_out/r06; mkdir -p $out
cd $root
bash scripts/_pmc_render.sh > /dev/null 2>&1
bash scripts/collect_profiles.sh r06 > $out/collect.log 2>&1
for i in 1 2 3 4 5; do python bench.py --extra r9_phone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['r9_phone']['frames_per_s'])"; done > $out/r06_r9_phone_five_processes.log
python bench.py --steps 20 --warmup 5 > $out/r06_bench_k20.json 2> $out/r06_bench_k20.err
ls $out | head -80
