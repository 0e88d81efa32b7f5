# usage: bash tools/run_big_dbg.sh <tag> "<env assignments>" ...   -- kernel stats of the 500-locus workload under debug switches
O=gpurun_out/$1; mkdir -p $O; shift
cd /tmp && export TMPDIR=/tmp
i=0
for envs in "$@"; do
  i=$((i+1))
  export $envs
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/p$i -o big -- python3 $GRAFT_REPO_ROOT/bench.py --workload big --steps 4 --warmup 1 --cpu-sample 0 --no-checks > /dev/null 2>&1
  echo "== $envs"; python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/$O/p$i/big_kernel_stats.csv | head -4
  for e in $envs; do unset ${e%%=*}; done
done
