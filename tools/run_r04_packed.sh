#!/bin/bash
# round 4: the headline workload with the batch in the 2-bit packed form beside the ASCII form -- bench lines (with e2e for both ingest
# formats), kernel stats, PMC traffic and SQ counters of the packed run
tag=${1:-p}
O=gpurun_out/r04/$tag; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_mtb.json 2> $O/bench_mtb.err
timeout 400 python bench.py --steps 20 --warmup 5 --input packed > $O/bench_mtb_packed.json 2> $O/bench_mtb_packed.err
tail -3 $O/bench_mtb_packed.err
cd /tmp && export TMPDIR=/tmp
for inp in ascii packed; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$inp -o mtb_$inp -- python3 $R/bench.py --input $inp --steps 5 --warmup 2 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$O/pmc_packed -o $c -- python3 $R/bench.py --input packed --steps 3 --warmup 1 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
done
n=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  n=$((n+1))
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$O/sq_packed -o p$n -- python3 $R/bench.py --input packed --steps 2 --warmup 1 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
done
cd $R
{
for inp in ascii packed; do echo "== mtb $inp"; python tools/kstats.py $O/prof_$inp/mtb_${inp}_kernel_stats.csv; done
echo "== pmc packed"; python tools/pmc_summary.py $O/pmc_packed; python tools/pmc_summary.py $O/sq_packed
} > $O/summary.txt 2>&1
for f in $O/prof_*/*_kernel_stats.csv; do python tools/trim_csv.py $f $O/$(basename $f) 2>/dev/null; done
for d in $O/pmc_packed $O/sq_packed; do for f in $d/*counter_collection.csv; do python tools/trim_csv.py $f $O/$(basename $d)_$(basename $f) 2>/dev/null; done; done
rm -rf $O/prof_* $O/pmc_packed/ $O/sq_packed/ 2>/dev/null
cat $O/summary.txt
python - <<PY
import json
for f in ("bench_mtb", "bench_mtb_packed"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "ms/step %.3f value %.3e kernel %.3f ms frac %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]), d["step_ms"], d["config"].get("full_size_packed_equals_ascii"), d["config"].get("full_size_shard_invariance"), {k: (round(v.get("seconds", 0), 4), v.get("coverage_equals_hbm_resident_run")) for k, v in d.get("e2e", {}).items() if isinstance(v, dict)})
    except Exception as e:
        print(f, "FAILED", e)
PY
