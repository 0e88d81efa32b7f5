#!/bin/bash
# read_cluster forms side by side: kernel stats of mtb / big / nanopore / mtb-x8 with the wave form first (default) and with the
# workgroup form alone (DRPRG_RC_FORM=wg), plus the bench lines
tag=${1:-rc}
O=gpurun_out/r04/$tag; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for wl in mtb big nanopore mtb-x8; do
  for form in wave wg; do
    DRPRG_RC_FORM=$form timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_${wl}_$form -o x -- python3 $R/bench.py --workload $wl --steps 5 --warmup 2 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
  done
done
cd $R
for wl in mtb big nanopore mtb-x8; do
  for form in wave wg; do
    echo "== $wl $form"; python tools/kstats.py $O/prof_${wl}_$form/x_kernel_stats.csv | grep -v rocclr
    python tools/trim_csv.py $O/prof_${wl}_$form/x_kernel_stats.csv $O/${wl}_${form}_kernel_stats.csv
  done
done > $O/summary.txt 2>&1
rm -rf $O/prof_*
cat $O/summary.txt
for wl in mtb big nanopore; do
  for form in wave wg; do
    DRPRG_RC_FORM=$form timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --cpu-sample 0 --e2e 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl $form', 'ms/step %.3f' % d['ms_per_step'], d['step_ms']['median'], d['config']['full_size_shard_invariance'], d['config']['full_size_direct_vs_filtered_kernel_identical'], d['config']['leftover_reads_per_batch'])"
  done
done
