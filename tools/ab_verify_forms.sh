#!/bin/bash
# the filtered sequence with verify_scan_kernel (default) against cand_scan + cand_gather + verify_count_kernel (DRPRG_VERIFY_FORM=gather)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05/abv; mkdir -p $O
for wl in ${@:-mtb mtb-dense mtb-x2 mtb-x8 nanopore}; do for form in default gather; do
  unset DRPRG_VERIFY_FORM; [ $form = gather ] && export DRPRG_VERIFY_FORM=gather
  st=20; [ $wl = nanopore ] && st=5
  timeout 400 python bench.py --workload $wl --steps $st --warmup 3 --cpu-sample 0 --e2e 0 --no-checks > $O/b.json 2> $O/b.err
  ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python3 $R/bench.py --workload $wl --steps 5 --warmup 2 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1 )
  echo "== $wl $form: step $(python -c "import json;print('%.3f ms' % json.loads(open('$O/b.json').read().strip().splitlines()[-1])['ms_per_step'])") | $(python $R/tools/kstats.py $O/prof/k_kernel_stats.csv | grep 'verify\|cand_' | awk '{print $1, $4}' | sed 's/drprg::dev:://' | tr '\n' ' ')"; rm -rf $O/prof
done; done
