# (every command under its own timeout: one call on a sluggish box once ran into gpurun's limit and cost 40 GPU-minutes)
# usage: bash tools/run_profiles.sh <tag>   -- everything profiles/<round>/ holds: bench lines, kernel stats, PMC traffic, SQ counters
O=gpurun_out/$1; mkdir -p $O
timeout 300 python bench.py > $O/bench_mtb.json 2> $O/bench_mtb.err
timeout 300 python bench.py --workload nanopore --steps 5 > $O/bench_nanopore.json 2> $O/bench_nanopore.err
timeout 300 python bench.py --workload big --steps 5 > $O/bench_big.json 2> $O/bench_big.err
DRPRG_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 5 --warmup 1 --cpu-sample 0 2> $O/bench_gloo2.err | grep '^{' > $O/bench_gloo2.json
cd /tmp && export TMPDIR=/tmp
for wl in mtb nanopore big; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$wl -o $wl -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 1 --cpu-sample 0 --no-checks > /dev/null 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$wl -o $c -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample 0 --no-checks > /dev/null 2>&1
  done
done
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  n=$((n+1))
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$O/sq_big -o p$n -- python3 $GRAFT_REPO_ROOT/bench.py --workload big --steps 2 --warmup 1 --cpu-sample 0 --no-checks > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
for wl in mtb nanopore big; do echo "== $wl"; python tools/kstats.py $O/prof_$wl/${wl}_kernel_stats.csv; python tools/pmc_summary.py $O/pmc_$wl; done > $O/summary.txt
echo "== big SQ" >> $O/summary.txt; python tools/pmc_summary.py $O/sq_big >> $O/summary.txt
cat $O/summary.txt
