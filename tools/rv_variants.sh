#!/bin/bash
# read_verify_kernel build variants (experimental libraries under build/<name>/): the kernel's time on the given workloads against the lane form
libs=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05/rvv; mkdir -p $O
for lib in $libs; do
  export DRPRG_HIP_LIB=$R/build/$lib/libdrprg_hip.so DRPRG_VERIFY_FORM=read
  [ "$lib" = lane ] && { export DRPRG_HIP_LIB=$R/build/exp/libdrprg_hip.so; export DRPRG_VERIFY_FORM=gather; }
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "read_by_read or short_reads_bit_exact or config1" > $O/t.txt 2>&1; echo "== $lib: $(tail -1 $O/t.txt)"
  for wl in ${@:-mtb}; do
    ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python3 $R/bench.py --workload $wl --steps 5 --warmup 2 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1 )
    echo "   $wl: $(python $R/tools/kstats.py $O/prof/k_kernel_stats.csv | grep 'verify' | tr -s ' ')"; rm -rf $O/prof
  done
done
