# usage: [PMC_BENCH_ARGS="--input packed"] bash tools/pmc_kernel.sh <tag> <kernel-substring> "<counters of pass 1>" ["<counters of pass 2>" ...]
cd /tmp && export TMPDIR=/tmp
tag=$1; kern=$2; shift; shift
n=0
for c in "$@"; do
  n=$((n+1))
  rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmck_$tag -o p$n -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-checks $PMC_BENCH_ARGS > /dev/null 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/pmck_$tag | grep "$kern"
