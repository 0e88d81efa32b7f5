cd /tmp && export TMPDIR=/tmp
for d in ${DBG_LIST:-0 16}; do
export DRPRG_FT_DEBUG=$d
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dbg$d -o x -- python3 $GRAFT_REPO_ROOT/bench.py --workload mtb --steps 5 --warmup 1 --cpu-sample 0 --no-checks --e2e 0 > /dev/null 2>&1 < /dev/null
f=$(find $GRAFT_REPO_ROOT/gpurun_out/dbg$d -name '*kernel_stats.csv' | head -1)
echo "== DRPRG_FT_DEBUG=$d"; [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/kstats.py "$f" < /dev/null | grep -i "${DBG_GREP:-verify\|read_cluster\|filter}" 
done
