# (every command under its own timeout: a sluggish box must not run a whole call into gpurun's limit)
# usage: bash tools/run_three.sh <tag>  -- GPU parity suite + kernel stats of the three workloads (read_cluster / verify tuning loop)
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for wl in mtb nanopore big; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$wl -o $wl -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 8 --warmup 1 --cpu-sample 0 --no-checks > $GRAFT_REPO_ROOT/$O/bench_$wl.json 2> /dev/null
  echo "== $wl $(python3 -c "import json;d=json.loads(open('$GRAFT_REPO_ROOT/$O/bench_$wl.json').read().strip().splitlines()[-1]);print('ms_per_step %.3f'%d['ms_per_step'])")"
  python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/$O/prof_$wl/${wl}_kernel_stats.csv | grep -E "read_cluster|verify|sketch|refine|gather" | head -6
done
