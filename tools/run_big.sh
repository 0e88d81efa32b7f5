# (every command under its own timeout: a sluggish box must not run a whole call into gpurun's limit)
# usage: bash tools/run_big.sh <tag>   -- parity subset + bench of the 500-locus workload + kernel stats
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config4 or short_reads or long_reads or ragged or randomized or several_groups or full_size or do_not_fit or longer_than" 2>&1 | tail -5 > $O/gputest.txt
timeout 300 python bench.py --workload big --steps 5 --cpu-sample 0 > $O/bench_big.json 2> $O/bench_big.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_big -o big -- python3 $GRAFT_REPO_ROOT/bench.py --workload big --steps 5 --warmup 1 --cpu-sample 0 --no-checks > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/kstats.py $O/prof_big/big_kernel_stats.csv > $O/kstats.txt
cat $O/gputest.txt; python -c "
import json,sys
d=json.loads(open('$O/bench_big.json').read().strip().splitlines()[-1])
print('ms_per_step',d['ms_per_step'],'value',d['value'],'kernel',d['roofline']['kernel'],d['roofline']['avg_launch_ms'],'checks',d['config']['full_size_shard_invariance'],d['config']['full_size_direct_vs_filtered_kernel_identical'])
"; tail -3 $O/bench_big.err; cat $O/kstats.txt
