#!/bin/bash
# Where verify_scan_kernel's time goes: DRPRG_FT_DEBUG=16 (no window test: every index k-mer inside its read counts as a minimizer) and
# =32 (no table probe and nothing after it) against the whole kernel; results are wrong under both, so --no-checks.   usage: ... wl...
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05/vabl; mkdir -p $O
for wl in ${@:-mtb}; do for dbg in ${V_DBG:-0 16 32 512 2048}; do
  export DRPRG_FT_DEBUG=$dbg
  ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python3 $R/bench.py --workload $wl --steps 5 --warmup 2 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1 )
  echo "$wl debug=$dbg: $(python $R/tools/kstats.py $O/prof/k_kernel_stats.csv | grep -E 'verify_|read_cluster' | sed -E 's/drprg::dev:://; s/min=.*//' | tr -s ' ' | tr '\n' ';')"; rm -rf $O/prof
done; done
