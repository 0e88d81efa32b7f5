#!/bin/bash
# crossover of the middle tier against the direct sequence (sketch_wave_kernel): bench line per workload with kernel 3 forced
tag=${1:-k3}; shift
wls=${@:-"mtb-x4 mtb-x8 mtb-x16"}
out=gpurun_out/r03; mkdir -p $out
for wl in $wls; do
  DRPRG_BENCH_KERNEL=3 timeout 200 python bench.py --workload $wl --steps 10 --warmup 3 --cpu-sample 0 --e2e 0 --no-checks > $out/${tag}_bench_${wl}.json 2> $out/${tag}_bench_${wl}.err
  python - <<PY
import json
try:
    d = json.loads(open("$out/${tag}_bench_${wl}.json").read().strip().splitlines()[-1])
    print("$wl kernel3", "ms/step %.3f" % d["ms_per_step"], d["roofline"]["kernel"], "%.3f ms" % d["roofline"]["avg_launch_ms"], "nodes", d["config"]["kmer_nodes"])
except Exception as e:
    print("$wl", "no line:", e)
PY
done
