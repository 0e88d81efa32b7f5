#!/bin/bash
# precise A/B of library builds: per workload, <rounds> alternations of bench.py with <steps> steps per library; prints the median step (and the
# packed leg's for mtb).   usage: tools/lib_steps.sh "<libs: default | path ...>" <rounds> <steps> wl...
libs=$1; rounds=$2; steps=$3; shift; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
for wl in ${@:-mtb}; do
  for r in $(seq 1 $rounds); do for lib in $libs; do
    unset DRPRG_HIP_LIB; [ "$lib" != default ] && export DRPRG_HIP_LIB=$R/$lib
    timeout 600 python bench.py --workload $wl --steps $steps --warmup 5 --cpu-sample 0 --e2e 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('packed_input') or {}
print('$wl %-28s step median %.4f mean %.4f' % ('$lib', d['step_ms']['median'], d['ms_per_step']), ('packed %.4f' % p['step_ms']['median']) if p else '')"
  done; done
done
