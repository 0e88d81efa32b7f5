O=gpurun_out/r02m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
wl=big
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$wl -o $wl -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 1 --cpu-sample 0 --no-checks --e2e 0 > /dev/null 2>&1 < /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$wl -o $c -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample 0 --no-checks --e2e 0 > /dev/null 2>&1 < /dev/null
done
n=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  n=$((n+1))
  timeout 150 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$O/sq_big -o p$n -- python3 $GRAFT_REPO_ROOT/bench.py --workload big --steps 2 --warmup 1 --cpu-sample 0 --no-checks --e2e 0 > /dev/null 2>&1 < /dev/null
done
cd $GRAFT_REPO_ROOT
echo "== big" > $O/summary.txt; python tools/kstats.py $O/prof_big/big_kernel_stats.csv >> $O/summary.txt; python tools/pmc_summary.py $O/pmc_big >> $O/summary.txt
echo "== big SQ" >> $O/summary.txt; python tools/pmc_summary.py $O/sq_big >> $O/summary.txt
head -8 $O/summary.txt
