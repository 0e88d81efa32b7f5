# read_cluster's fixed cost per launch against batch size: `bash tools/rw_debug.sh [wave|wg]` on the GPU box (default: the workgroup form)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DRPRG_RC_FORM=${1:-wg}
for n in 100000 1000000 10000000; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04/dbg/p$n -o x -- python3 $R/bench.py --reads-per-gpu $n --steps 5 --warmup 2 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
  echo "reads $n"; python3 $R/tools/kstats.py $R/gpurun_out/r04/dbg/p$n/x_kernel_stats.csv | grep -E "read_cluster|rw_totals|verify|hit_scan|cand_"
done
rm -rf $R/gpurun_out/r04/dbg
