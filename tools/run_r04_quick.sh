#!/bin/bash
# edit-measure loop of the filter kernel: a parity subset, then the ascii and packed bench lines + kernel stats
tag=${1:-q}; sel=${2:-"short_reads or ragged or dense or config1 or packed or second_stage or middle_tier or scaled"}
O=gpurun_out/r04/$tag; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
timeout 1200 python -m pytest tests -x -q -m gpu -k "$sel" > $O/gputests.txt 2>&1; tail -4 $O/gputests.txt
for inp in ascii packed; do
  timeout 300 python bench.py --steps 20 --warmup 5 --input $inp --cpu-sample 0 --e2e 0 > $O/bench_$inp.json 2> $O/bench_$inp.err
done
cd /tmp && export TMPDIR=/tmp
for inp in ascii packed; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$inp -o mtb_$inp -- python3 $R/bench.py --input $inp --steps 5 --warmup 2 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
done
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/$O/sq_packed -o p1 -- python3 $R/bench.py --input packed --steps 2 --warmup 1 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
cd $R
for inp in ascii packed; do echo "== mtb $inp"; python tools/kstats.py $O/prof_$inp/mtb_${inp}_kernel_stats.csv | grep -v rocclr; done
python tools/pmc_summary.py $O/sq_packed | grep sketch_filter
rm -rf $O/prof_* $O/sq_packed
python - <<PY
import json
for f in ("bench_ascii", "bench_packed"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "ms/step %.3f value %.3e kernel %.3f ms frac %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]), d["step_ms"]["median"], d["config"].get("full_size_packed_equals_ascii"), d["config"].get("full_size_shard_invariance"), d["config"].get("full_size_direct_vs_filtered_kernel_identical"))
    except Exception as e:
        print(f, "FAILED", e)
PY
