#!/bin/bash
# round 5 edit-measure loop: a parity subset (stops at the first failure), then the ascii (and packed) bench lines + kernel stats
#   tools/run_r05_quick.sh <tag> "<pytest -k>" [workloads...]     
tag=${1:-q}; sel=${2:-"short_reads or ragged or dense or config1 or packed or second_stage or middle_tier or scaled"}; shift; shift
wls=${@:-mtb}
O=gpurun_out/r05/$tag; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
if [ -n "$sel" ] && [ "$sel" != "none" ]; then
  timeout 1500 python -m pytest tests -x -q -m gpu -k "$sel" > $O/gputests.txt 2>&1; tail -4 $O/gputests.txt
  grep -q " passed" $O/gputests.txt && ! grep -q "failed\|error" $O/gputests.txt || { tail -60 $O/gputests.txt; exit 1; }
fi
forms="default"; [ -n "$VERIFY_AB" ] && forms="default gather"   # gather: cand_scan + cand_gather + verify_count_kernel (rounds 1-4)
for wl in $wls; do for form in $forms; do for inp in ascii packed; do
  [ "$wl" != "mtb" ] && [ "$inp" = "packed" ] && continue
  unset DRPRG_VERIFY_FORM; [ "$form" = gather ] && export DRPRG_VERIFY_FORM=gather
  timeout 400 python bench.py --workload $wl --steps 20 --warmup 5 --input $inp --cpu-sample 0 --e2e 0 > $O/bench_${wl}_${form}_$inp.json 2> $O/bench_${wl}_${form}_$inp.err
  ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o k -- python3 $R/bench.py --workload $wl --input $inp --steps 5 --warmup 2 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1 )
  echo "== $wl $form $inp"; python tools/kstats.py $O/prof/k_kernel_stats.csv | grep -v rocclr
  cp $O/prof/k_kernel_stats.csv $O/kstats_${wl}_${form}_$inp.csv 2>/dev/null; rm -rf $O/prof
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_${wl}_${form}_$inp.json").read().strip().splitlines()[-1])
    print("ms/step %.3f value %.3e kernel %.3f ms frac %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]), d["step_ms"]["median"], {k: v for k, v in d["config"].items() if k.startswith("full_size")})
except Exception as e:
    print("FAILED", e); print(open("$O/bench_${wl}_${form}_$inp.err").read()[-1500:])
PY
done; done; done
