# (every command under its own timeout: a sluggish box must not run a whole call into gpurun's limit)
# usage: bash tools/run_mtb.sh <tag> ["ENV=.. ENV=.."]   -- parity subset + bench + kernel stats of the headline workload
O=gpurun_out/$1; mkdir -p $O
[ -n "$2" ] && export $2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config1 or config2 or short_reads or ragged or randomized or several_groups" 2>&1 | tail -3 > $O/gputest.txt
timeout 300 python bench.py --steps 20 --cpu-sample 0 > $O/bench_mtb.json 2> $O/bench_mtb.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o mtb -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 1 --cpu-sample 0 --no-checks > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/kstats.py $O/prof/mtb_kernel_stats.csv > $O/kstats.txt
cat $O/gputest.txt; python -c "
import json
d=json.loads(open('$O/bench_mtb.json').read().strip().splitlines()[-1])
print('ms_per_step',d['ms_per_step'],'value',d['value'],'kernel',d['roofline']['kernel'],d['roofline']['avg_launch_ms'],'frac',d['roofline']['frac'],'checks',d['config']['full_size_shard_invariance'],d['config']['full_size_direct_vs_filtered_kernel_identical'])
"; tail -2 $O/bench_mtb.err; head -9 $O/kstats.txt
