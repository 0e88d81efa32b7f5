#!/bin/bash
# verification kernel A/B on several workloads: lane form (the default) against the read form (DRPRG_VERIFY_FORM=read)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05/ab; mkdir -p $O
for wl in ${@:-mtb mtb-x2 mtb-x8}; do for v in lane read; do
  export DRPRG_VERIFY_FORM=read DRPRG_FT_DEBUG=0
  [ $v = lane ] && unset DRPRG_VERIFY_FORM   # (the default: one lane per candidate, verify_scan_kernel)
  ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python3 $R/bench.py --workload $wl --steps 5 --warmup 2 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1 )
  echo "== $wl $v: $(python $R/tools/kstats.py $O/prof/k_kernel_stats.csv | grep 'verify' | tr -s ' ')"; rm -rf $O/prof
done; done
