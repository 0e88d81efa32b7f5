# usage: bash tools/pmc.sh <tag> <workload> <COUNTER> [<COUNTER> ...]     (one rocprofv3 --pmc pass per counter)
# every launch of the run is a timed full-size one (--no-checks), so per-launch = sum / launches
cd /tmp && export TMPDIR=/tmp
tag=$1; wl=$2; shift; shift
for c in "$@"; do
  rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o $c -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample 0 --no-checks > /dev/null 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
