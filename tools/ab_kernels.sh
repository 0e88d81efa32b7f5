#!/bin/bash
# same-box A/B of two builds: per-kernel averages under rocprofv3 (warm device: the bench's spin-up launches are in the averages), alternating
#   bash tools/ab_kernels.sh <lib A> <lib B> [workload] [bench args]
A=$1; B=$2; wl=${3:-mtb}; shift; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/ab; mkdir -p $O
for i in 1 2; do for l in $A $B; do
  ( cd /tmp && export TMPDIR=/tmp && DRPRG_HIP_LIB=$R/$l timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python3 $R/bench.py --workload $wl --steps 20 --warmup 5 --cpu-sample 0 --e2e 0 --no-checks "$@" > /dev/null 2>&1 )
  echo "== $l"; python $R/tools/kstats.py $O/prof/k_kernel_trace.csv | grep "drprg::dev" | grep -v counters_home | sed -E 's/drprg::dev:://' | tr -s ' ' | cut -c1-110; rm -rf $O/prof
done; done
