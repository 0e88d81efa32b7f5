#!/bin/bash
# round 6 edit-measure loop: a parity subset (stops at the first failure), then bench lines + kernel stats of the given workloads under the given
# environment variants (same box, back to back)
#   tools/run_r06_quick.sh <tag> "<pytest -k | none>" "<workloads>" "<name:ENV=v@ENV=v;name2:...>" [inputs]
tag=${1:-q}; sel=${2:-"short_reads or ragged or dense or config1 or packed or second_stage or middle_tier or scaled"}
wls=${3:-mtb}; variants=${4:-"default:"}; inputs=${5:-"ascii packed"}
O=gpurun_out/r06/$tag; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
if [ -n "$sel" ] && [ "$sel" != "none" ]; then
  timeout 2400 python -m pytest tests -x -q -m gpu -k "$sel" > $O/gputests.txt 2>&1; tail -4 $O/gputests.txt
  grep -q " passed" $O/gputests.txt && ! grep -q "failed\|error" $O/gputests.txt || { tail -80 $O/gputests.txt; exit 1; }
fi
IFS=';' read -ra VARS <<< "$variants"
for wl in $wls; do for inp in $inputs; do
  [ "$wl" != "mtb" ] && [ "$inp" = "packed" ] && continue
  for v in "${VARS[@]}"; do
    name=${v%%:*}; envs=${v#*:}
    ( IFS="@"; for e in $envs; do [ -n "$e" ] && export "$e"; done; unset IFS
      timeout 400 python bench.py --workload $wl --steps ${STEPS:-20} --warmup ${WARMUP:-5} --input $inp --cpu-sample 0 --e2e 0 > $O/bench_${wl}_${name}_$inp.json 2> $O/bench_${wl}_${name}_$inp.err
      if [ -z "$NOPROF" ]; then
        ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o k -- python3 $R/bench.py --workload $wl --input $inp --steps 20 --warmup 5 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1 )
        echo "== $wl $name $inp"; python tools/kstats.py $O/prof/k_kernel_stats.csv | grep -v rocclr
        cp $O/prof/k_kernel_stats.csv $O/kstats_${wl}_${name}_$inp.csv 2>/dev/null; rm -rf $O/prof
      else echo "== $wl $name $inp"; fi
      python - <<PY
import json
try:
    d = json.loads(open("$O/bench_${wl}_${name}_$inp.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("ms/step %.4f value %.3e kernel %.4f ms frac %.3f" % (d["ms_per_step"], d["value"], r["avg_launch_ms"], r["frac"]), d["step_ms"], r.get("filter_schedule"), {k: v for k, v in d["config"].items() if k.startswith("full_size")})
except Exception as e:
    print("FAILED", e); print(open("$O/bench_${wl}_${name}_$inp.err").read()[-1500:])
PY
    )
  done
done; done
