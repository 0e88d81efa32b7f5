#!/bin/bash
# whole GPU suite + the headline bench line (one gpurun call); everything under its own timeout
tag=${1:-full}
out=gpurun_out/r03; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu > $out/${tag}_gputests.txt 2>&1; tail -5 $out/${tag}_gputests.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_mtb.json 2> $out/${tag}_bench_mtb.err; tail -2 $out/${tag}_bench_mtb.err
python - <<PY
import json
d = json.loads(open("$out/${tag}_bench_mtb.json").read().strip().splitlines()[-1])
print("mtb ms/step %.3f value %.3e kernel %.3f ms frac %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]), d["step_ms"], d.get("cpu_baseline", {}).get("parity_vs_hip_on_sample"), {k: v.get("reads_per_s") for k, v in d.get("e2e", {}).items() if isinstance(v, dict)})
PY
