#!/bin/bash
# read_cluster_kernel build variants (Makefile EXTRA_DEFS: slots per thread / staged hits / workgroups per CU), each a library of its own
# under build/<name>/: a parity subset, then the kernel's time on the given workloads.   usage: tools/rc_variants.sh "<libs>" wl...
libs=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05/rcv; mkdir -p $O
for lib in $libs; do
  export DRPRG_HIP_LIB=$R/$lib
  [ "$lib" = default ] && unset DRPRG_HIP_LIB
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "short_reads or several_groups or do_not_fit or longer_than or config1 or config4" > $O/t.txt 2>&1; echo "== $lib: $(tail -1 $O/t.txt)"
  for wl in ${@:-mtb}; do
    ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python3 $R/bench.py --workload $wl --steps 5 --warmup 2 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1 )
    echo "   $wl: $(python $R/tools/kstats.py $O/prof/k_kernel_stats.csv | grep 'read_cluster' | tr -s ' ')"; rm -rf $O/prof
  done
done
