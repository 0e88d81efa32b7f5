# (every command under its own timeout: a sluggish box must not run a whole call into gpurun's limit)
set -x
O=gpurun_out/r02a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/gputest.txt
timeout 300 python bench.py > $O/bench_mtb.json 2> $O/bench_mtb.err
timeout 300 python bench.py --workload mtb-random --cpu-sample 0 > $O/bench_mtb_random.json 2>> $O/bench_mtb.err
timeout 300 python bench.py --workload nanopore --steps 5 > $O/bench_nanopore.json 2> $O/bench_nanopore.err
timeout 300 python bench.py --workload big --steps 5 > $O/bench_big.json 2> $O/bench_big.err
DRPRG_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 5 --warmup 1 --cpu-sample 0 > $O/bench_gloo2.json 2> $O/bench_gloo2.err
timeout 300 python bench.py --gpus 2 --steps 2 > $O/bench_refuse.txt 2>&1; echo "rc=$?" >> $O/bench_refuse.txt
cd /tmp && export TMPDIR=/tmp
for wl in mtb nanopore big; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$wl -o $wl -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 1 --cpu-sample 0 --no-checks > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
for wl in mtb nanopore big; do echo "== $wl"; python tools/kstats.py $(find $O/prof_$wl -name "*kernel_stats.csv" | head -1); done > $O/kstats.txt
cat $O/gputest.txt $O/bench_*.json $O/bench_refuse.txt $O/kstats.txt
