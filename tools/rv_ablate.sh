#!/bin/bash
# read_verify_kernel taken apart: its time with parts switched off (wrong results, timing only)
O=gpurun_out/r05/ablate; mkdir -p $O; R=${GRAFT_REPO_ROOT:-$PWD}
for wl in ${@:-mtb}; do for dbg in 0 64 128 192 448; do
  ( cd /tmp && export TMPDIR=/tmp DRPRG_VERIFY_FORM=${DRPRG_VERIFY_FORM:-read} DRPRG_FT_DEBUG=$dbg && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o k -- python3 $R/bench.py --workload $wl --steps 5 --warmup 2 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1 )
  echo "== $wl debug $dbg: $(python tools/kstats.py $O/prof/k_kernel_stats.csv | grep read_verify)"; rm -rf $O/prof
done; done
