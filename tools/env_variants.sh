#!/bin/bash
# A/B of one environment switch of the library: per value a parity subset, the post-filter kernels' averages under rocprofv3, bench step times.
#   usage: tools/env_variants.sh NAME "<values; - = unset>" wl...
name=$1; vals=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05/envv; mkdir -p $O
for v in $vals; do
  unset $name; [ "$v" != "-" ] && export $name=$v
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${SEL:-short_reads or ragged or dense or several_groups or do_not_fit or longer_than or config1 or packed or candidates_at_both_ends}" > $O/t.txt 2>&1; echo "== $name=$v: $(tail -1 $O/t.txt)"
  grep -q "failed\|error" $O/t.txt && tail -40 $O/t.txt
  for wl in ${@:-mtb}; do for inp in ascii packed; do
    [ "$wl" != "mtb" ] && [ "$inp" = "packed" ] && continue
    ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python3 $R/bench.py --workload $wl --input $inp --steps 5 --warmup 2 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1 )
    echo "   $wl $inp: $(python $R/tools/kstats.py $O/prof/k_kernel_stats.csv | grep -E 'verify_|read_cluster' | sed -E 's/drprg::dev:://; s/min=.*//' | tr -s ' ' | tr '\n' ';')"; rm -rf $O/prof
    for rep in 1 2; do
      timeout 400 python bench.py --workload $wl --steps 20 --warmup 5 --input $inp --cpu-sample 0 --e2e 0 > $O/b.json 2> $O/b.err
      python - <<PY
import json
try:
    d = json.loads(open("$O/b.json").read().strip().splitlines()[-1])
    print("      ms/step %.4f median %.4f filter kernel %.3f ms" % (d["ms_per_step"], d["step_ms"]["median"], d["roofline"]["avg_launch_ms"]))
except Exception as e:
    print("FAILED", e); print(open("$O/b.err").read()[-1500:])
PY
    done
  done; done
done
