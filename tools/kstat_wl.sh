# per-kernel averages of one workload: bash tools/kstat_wl.sh <workload> [ENV=..]
wl=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/ks_$wl; rm -rf $out
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out -o x -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 1 --cpu-sample 0 --no-checks --e2e 0 > /dev/null 2>&1 < /dev/null
f=$(find $out -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/kstats.py "$f" < /dev/null | grep "drprg::" | head -12
