#!/bin/bash
# One gpurun call: middle-tier parity tests, then a bench line + kernel stats per index size (8d index grown 1-32 fold).
#   tools/run_scaled.sh <tag> [workloads...]      -> gpurun_out/r03/<tag>_*
tag=${1:-s}; shift
wls=${@:-"mtb mtb-x2 mtb-x4 mtb-x8 mtb-x16"}
out=gpurun_out/r03
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
if [ -z "$SKIP_TESTS" ]; then
  timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "middle_tier or scaled" > $out/${tag}_tests.txt 2>&1
  tail -3 $out/${tag}_tests.txt
  grep -q " passed" $out/${tag}_tests.txt && ! grep -q "failed\|error" $out/${tag}_tests.txt || { echo "tests not green: stopping"; tail -30 $out/${tag}_tests.txt; exit 1; }
fi
for wl in $wls; do
  DRPRG_FT_STATS=1 timeout 200 python bench.py --workload $wl --steps 10 --warmup 3 --cpu-sample 0 --e2e 0 > $out/${tag}_bench_${wl}.json 2> $out/${tag}_bench_${wl}.err
  tail -2 $out/${tag}_bench_${wl}.err
  python - <<PY
import json
try:
    d = json.loads(open("$out/${tag}_bench_${wl}.json").read().strip().splitlines()[-1])
    print("$wl", "ms/step %.3f" % d["ms_per_step"], "kernel %.3f ms" % d["roofline"]["avg_launch_ms"], "nodes", d["config"]["kmer_nodes"], "hits", d["config"]["hits_per_batch"],
          "checks", d["config"]["full_size_shard_invariance"], d["config"]["full_size_direct_vs_filtered_kernel_iden" if "full_size_direct_vs_filtered_kernel_iden" in d["config"] else "full_size_direct_vs_filtered_kernel_identical"])
except Exception as e:
    print("$wl", "no line:", e)
PY
  R=${GRAFT_REPO_ROOT:-$PWD}
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_${tag}_${wl} -o p -- python3 $R/bench.py --workload $wl --steps 10 --warmup 3 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1)
  f=$(find $out/prof_${tag}_${wl} -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python tools/kstats.py $f > $out/${tag}_kstats_${wl}.txt 2>&1 && head -14 $out/${tag}_kstats_${wl}.txt
  rm -rf $out/prof_${tag}_${wl}
done
