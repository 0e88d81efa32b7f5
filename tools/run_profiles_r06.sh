#!/bin/bash
# everything profiles/r06/ holds of the final code, in one gpurun call (every command under its own timeout); parts by name so that a call
# can be split: tools/run_profiles_r06.sh <tag> [bench] [ranks] [series] [trace] [driver] [pmc] [sq]   (default: all)
#   bench : bench lines of mtb (defaults: 100 steps, cpu_baseline, e2e; and the driver's command --steps 20 --warmup 5), packed, nanopore (both formats), big,
#           and the second stage's other home for each format (DRPRG_FILTER_STAGE2)
#   ranks : bench.py --gpus 2 / 8 on one GPU over gloo, both --comm modes (control flow only: unmeasured on > 1 GPU)
#   series: bench lines of the index-size series
#   trace : rocprofv3 --kernel-trace --stats of every workload (per-dispatch rows kept: tools/kstats.py full-launch averages)
#   driver: the driver's command (--gpus 1 --steps 20 --warmup 5) under rocprofv3 --kernel-trace --stats
#   pmc   : FETCH_SIZE / WRITE_SIZE passes of mtb (ASCII and packed), mtb-x8, nanopore, big
#   sq    : SQ counters of mtb (ASCII and packed), mtb-x8, big
# The summary is written LAST, from the csv files it sits beside (their call counts are in it).
tag=${1:-z}; shift
parts=${@:-bench ranks series trace driver pmc sq}
has() { [[ " $parts " == *" $1 "* ]]; }
O=gpurun_out/r06/$tag; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
SERIES="mtb-dense mtb-x2 mtb-x4 mtb-x8 mtb-x16 mtb-x32"
if has bench; then
  timeout 500 python bench.py > $O/bench_mtb.json 2> $O/bench_mtb.err
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_mtb_driver_command.json 2> $O/bench_mtb_driver_command.err
  timeout 400 python bench.py --input packed --cpu-sample 0 --e2e 0 > $O/bench_mtb_packed.json 2> $O/bench_mtb_packed.err
  timeout 400 python bench.py --workload nanopore --steps 10 --warmup 3 --e2e 0 > $O/bench_nanopore.json 2> $O/bench_nanopore.err
  timeout 400 python bench.py --workload nanopore --input packed --steps 10 --warmup 3 --cpu-sample 0 --e2e 0 > $O/bench_nanopore_packed.json 2> $O/bench_nanopore_packed.err
  DRPRG_FILTER_STAGE2=lds timeout 400 python bench.py --input packed --steps 20 --warmup 5 --cpu-sample 0 --e2e 0 > $O/bench_mtb_packed_all_lds_second_stage.json 2> /dev/null
  DRPRG_FILTER_STAGE2=l2 timeout 400 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --e2e 0 > $O/bench_mtb_l2_second_stage.json 2> /dev/null
  timeout 400 python bench.py --workload big --steps 10 --warmup 3 --e2e 0 > $O/bench_big.json 2> $O/bench_big.err
  DRPRG_FT_SCHED=static timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --e2e 0 > $O/bench_mtb_static_schedule.json 2> $O/bench_mtb_static_schedule.err
fi
if has ranks; then
  for comm in native torch; do for n in 2 8; do
    DRPRG_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus $n --steps 5 --warmup 1 --cpu-sample 0 --comm $comm 2> $O/bench_gloo${n}_$comm.err | grep '^{' > $O/bench_${comm}_gloo_${n}ranks_one_gpu.json
  done; done
fi
if has series; then
  for wl in $SERIES; do
    timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --cpu-sample 0 --e2e 0 > $O/bench_$wl.json 2> $O/bench_$wl.err
    DRPRG_FT_SCHED=static timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --cpu-sample 0 --e2e 0 --no-checks > $O/bench_${wl}_static_schedule.json 2> /dev/null
  done
fi
cd /tmp && export TMPDIR=/tmp
if has trace; then
  for wl in mtb $SERIES nanopore big; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$wl -o $wl -- python3 $R/bench.py --workload $wl --steps 40 --warmup 10 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
  done
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_mtb-packed -o mtb-packed -- python3 $R/bench.py --input packed --steps 40 --warmup 10 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
fi
if has driver; then # the driver's own command under the profiler: the line's avg_launch_ms (HIP events) beside the profiler's table of the same process
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_mtb_under_rocprofv3 -o mtb_under_rocprofv3 -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/$O/bench_mtb_under_rocprofv3.json 2> /dev/null
fi
if has pmc; then
  for wl in mtb mtb-x8 nanopore big; do
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$O/pmc_$wl -o $c -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --spinup-ms 0 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
    done
  done
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$O/pmc_mtb-packed -o $c -- python3 $R/bench.py --input packed --steps 3 --warmup 1 --spinup-ms 0 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
  done
fi
if has sq; then
  for wl in mtb mtb-x8 big; do
    n=0
    for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
      n=$((n+1))
      timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$O/sq_$wl -o p$n -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --spinup-ms 0 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
    done
  done
  n=0
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    n=$((n+1))
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$O/sq_mtb-packed -o p$n -- python3 $R/bench.py --input packed --steps 2 --warmup 1 --spinup-ms 0 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
  done
fi
cd $R
# keep the csv rows of this project's kernels only (what goes into profiles/): stats, per-dispatch traces, counters
for f in $O/prof_*/*_kernel_stats.csv $O/prof_*/*_kernel_trace.csv; do [ -f "$f" ] && python tools/trim_csv.py $f $O/$(basename $f) 2>/dev/null; done
for d in $O/pmc_* $O/sq_*; do [ -d "$d" ] && for f in $d/*counter_collection.csv; do python tools/trim_csv.py $f $O/$(basename $d)_$(basename $f) 2>/dev/null; done; done
rm -rf $O/prof_* $O/pmc_*/ $O/sq_*/ 2>/dev/null
# the summary, from the trimmed files themselves
python tools/profile_summary.py $O > $O/summary.txt 2>&1
tail -80 $O/summary.txt
