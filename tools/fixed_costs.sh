#!/bin/bash
# what a launch costs before it does anything: the kernels' durations on batches of the given sizes (default 100 k / 1 M / 10 M reads; tools/README.md's table of round 5 used 20 k / 100 k / 1 M) (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05/fixed; mkdir -p $O
for n in ${@:-100000 1000000 10000000}; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python3 $R/bench.py --reads-per-gpu $n --input ${INP:-ascii} --steps 20 --warmup 3 --spinup-ms 0 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1 )
  echo "== $n reads"; python $R/tools/kstats.py $O/prof/k_kernel_stats.csv | grep "drprg::dev" | sed -E 's/drprg::dev:://' | tr -s ' '; rm -rf $O/prof
done
