#!/bin/bash
# everything profiles/r03/ holds, in one gpurun call (every command under its own timeout):
#   bench lines (full, with cpu_baseline + e2e) of mtb / nanopore / big; bench lines + kernel stats of the index-size series;
#   FETCH_SIZE / WRITE_SIZE passes of mtb, mtb-x2, mtb-x8, nanopore, big; SQ counters of the dominant kernel of mtb, mtb-x8 and big
# usage: bash tools/run_profiles_r03.sh <tag>
tag=${1:-h}
O=gpurun_out/r03/$tag; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_mtb.json 2> $O/bench_mtb.err
timeout 400 python bench.py --workload nanopore --steps 5 > $O/bench_nanopore.json 2> $O/bench_nanopore.err
timeout 400 python bench.py --workload big --steps 5 > $O/bench_big.json 2> $O/bench_big.err
DRPRG_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 5 --warmup 1 --cpu-sample 0 2> $O/bench_gloo2.err | grep '^{' > $O/bench_gloo_2ranks_one_gpu.json
for wl in mtb-dense mtb-x2 mtb-x4 mtb-x8 mtb-x16 mtb-x32; do
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --cpu-sample 0 --e2e 0 > $O/bench_$wl.json 2> $O/bench_$wl.err
done
DRPRG_BENCH_KERNEL=3 timeout 300 python bench.py --workload mtb-x8 --steps 10 --warmup 3 --cpu-sample 0 --e2e 0 --no-checks > $O/bench_mtb-x8_direct.json 2> /dev/null
DRPRG_MID_MAX_RECORDS=1000000 timeout 300 python bench.py --workload mtb-x16 --steps 10 --warmup 3 --cpu-sample 0 --e2e 0 --no-checks > $O/bench_mtb-x16_middle_tier.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
for wl in mtb mtb-dense mtb-x2 mtb-x4 mtb-x8 mtb-x16 mtb-x32 nanopore big; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$wl -o $wl -- python3 $R/bench.py --workload $wl --steps 5 --warmup 2 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
done
for wl in mtb mtb-x2 mtb-x8 nanopore big; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$O/pmc_$wl -o $c -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
  done
done
for wl in mtb mtb-x8 big; do
  n=0
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    n=$((n+1))
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$O/sq_$wl -o p$n -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
  done
done
cd $R
{
for wl in mtb mtb-dense mtb-x2 mtb-x4 mtb-x8 mtb-x16 mtb-x32 nanopore big; do echo "== $wl"; python tools/kstats.py $O/prof_$wl/${wl}_kernel_stats.csv; [ -d $O/pmc_$wl ] && python tools/pmc_summary.py $O/pmc_$wl; [ -d $O/sq_$wl ] && python tools/pmc_summary.py $O/sq_$wl; done
} > $O/summary.txt 2>&1
# keep the csv rows of this project's kernels only (what goes into profiles/)
for f in $O/prof_*/*_kernel_stats.csv; do python tools/trim_csv.py $f $O/$(basename $f) 2>/dev/null; done
for d in $O/pmc_* $O/sq_*; do for f in $d/*counter_collection.csv; do python tools/trim_csv.py $f $O/$(basename $d)_$(basename $f) 2>/dev/null; done; done
rm -rf $O/prof_* $O/pmc_*/ $O/sq_*/ 2>/dev/null
tail -80 $O/summary.txt
