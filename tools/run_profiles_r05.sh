#!/bin/bash
# everything profiles/r05/ holds of the final code, in one gpurun call (every command under its own timeout):
#   bench lines (full, with cpu_baseline + e2e) of mtb in both input formats, nanopore, big; bench lines + kernel stats of the
#   index-size series; bench.py --gpus 2 and --gpus 8 on one GPU over gloo for both --comm modes; FETCH_SIZE / WRITE_SIZE passes of mtb (ASCII and
#   packed), mtb-x8, nanopore, big; SQ counters of the dominant kernel of mtb (ASCII and packed) and big
# usage: bash tools/run_profiles_r05.sh <tag>
tag=${1:-z}
O=gpurun_out/r05/$tag; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
timeout 400 python bench.py > $O/bench_mtb.json 2> $O/bench_mtb.err
timeout 400 python bench.py --input packed > $O/bench_mtb_packed.json 2> $O/bench_mtb_packed.err
timeout 400 python bench.py --workload nanopore --steps 5 > $O/bench_nanopore.json 2> $O/bench_nanopore.err
timeout 400 python bench.py --workload big --steps 5 > $O/bench_big.json 2> $O/bench_big.err
for comm in native torch; do for n in 2 8; do
  DRPRG_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus $n --steps 5 --warmup 1 --cpu-sample 0 --comm $comm 2> $O/bench_gloo${n}_$comm.err | grep '^{' > $O/bench_${comm}_gloo_${n}ranks_one_gpu.json
done; done
for wl in mtb-dense mtb-x2 mtb-x4 mtb-x8 mtb-x16 mtb-x32; do
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --cpu-sample 0 --e2e 0 > $O/bench_$wl.json 2> $O/bench_$wl.err
done
cd /tmp && export TMPDIR=/tmp
for wl in mtb mtb-dense mtb-x2 mtb-x4 mtb-x8 mtb-x16 mtb-x32 nanopore big; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$wl -o $wl -- python3 $R/bench.py --workload $wl --steps 60 --warmup 10 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_mtb-packed -o mtb-packed -- python3 $R/bench.py --input packed --steps 60 --warmup 10 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
for wl in mtb mtb-x8 nanopore big; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$O/pmc_$wl -o $c -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
  done
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$O/pmc_mtb-packed -o $c -- python3 $R/bench.py --input packed --steps 3 --warmup 1 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
done
for wl in mtb big; do
  n=0
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    n=$((n+1))
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$O/sq_$wl -o p$n -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
  done
done
n=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  n=$((n+1))
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$O/sq_mtb-packed -o p$n -- python3 $R/bench.py --input packed --steps 2 --warmup 1 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
done
cd $R
{
for wl in mtb mtb-packed mtb-dense mtb-x2 mtb-x4 mtb-x8 mtb-x16 mtb-x32 nanopore big; do echo "== $wl"; python tools/kstats.py $O/prof_$wl/${wl}_kernel_stats.csv; [ -d $O/pmc_$wl ] && python tools/pmc_summary.py $O/pmc_$wl; [ -d $O/sq_$wl ] && python tools/pmc_summary.py $O/sq_$wl; done
} > $O/summary.txt 2>&1
# keep the csv rows of this project's kernels only (what goes into profiles/)
for f in $O/prof_*/*_kernel_stats.csv; do python tools/trim_csv.py $f $O/$(basename $f) 2>/dev/null; done
for d in $O/pmc_* $O/sq_*; do for f in $d/*counter_collection.csv; do python tools/trim_csv.py $f $O/$(basename $d)_$(basename $f) 2>/dev/null; done; done
rm -rf $O/prof_* $O/pmc_*/ $O/sq_*/ 2>/dev/null
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print("%-44s ms/step %.3f value %.3e kernel %.3f ms frac %.3f" % (os.path.basename(f), d["ms_per_step"], d["value"], r["avg_launch_ms"], r["frac"]),
              d["config"].get("input_format"), d["config"].get("comm"), {k: round(v.get("seconds", 0), 4) for k, v in d.get("e2e", {}).items() if isinstance(v, dict)})
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
tail -60 $O/summary.txt
