# one bounded rocprofv3 --kernel-trace --stats run of bench.py: bash tools/prof_one.sh <workload> <seconds>
wl=$1; lim=${2:-150}
out=$GRAFT_REPO_ROOT/gpurun_out/one_$wl; rm -rf $out
cd /tmp && export TMPDIR=/tmp
timeout $lim rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $wl -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 1 --cpu-sample 0 --no-checks --e2e 0 > /tmp/one.log 2>&1 < /dev/null
echo "rc=$? for $wl"; tail -2 /tmp/one.log | cut -c1-200
f=$(find $out -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/kstats.py "$f" < /dev/null | grep "drprg::" | head -8
