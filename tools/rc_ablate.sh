cd /tmp && export TMPDIR=/tmp
for d in 128 256 64 512 1024 48 0; do
  DRPRG_FT_DEBUG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_abl -o d$d -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-checks > /dev/null 2>&1
  echo "debug=$d"; python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/prof_abl/d${d}_kernel_stats.csv | grep read_cluster
done
