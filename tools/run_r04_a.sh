#!/bin/bash
# round 4, first GPU call: whole GPU suite on the refactored vectors / reduce, the issue-rate microbenchmark, the headline line,
# the N = 2 control flow of bench.py on one GPU for both --comm modes
tag=${1:-a}
O=gpurun_out/r04/$tag; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputests.txt 2>&1; tail -5 $O/gputests.txt
timeout 300 build/tools/mb_issue > $O/mb_issue.txt 2>&1; head -30 $O/mb_issue.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_mtb.json 2> $O/bench_mtb.err; tail -2 $O/bench_mtb.err
for comm in native torch; do
  DRPRG_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 5 --warmup 1 --cpu-sample 0 --comm $comm 2> $O/bench_gloo2_$comm.err | grep '^{' > $O/bench_${comm}_gloo_2ranks_one_gpu.json
  tail -3 $O/bench_gloo2_$comm.err
done
python - <<PY
import json
for f in ("bench_mtb", "bench_native_gloo_2ranks_one_gpu", "bench_torch_gloo_2ranks_one_gpu"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "ms/step %.3f value %.3e kernel %.3f ms frac %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]), d["config"]["workload"][:60], d["config"].get("comm"), d["config"].get("all_ranks_hold_the_sum_of_the_ranks_vectors"))
    except Exception as e:
        print(f, "FAILED", e)
PY
