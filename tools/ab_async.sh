# same-box A/B of the deferred (default) and the synchronous step loop of bench.py
for i in 1 2 3; do
  for m in 0 1; do
    DRPRG_BENCH_SYNC=$m timeout 200 python bench.py --cpu-sample 0 --e2e 0 --no-checks --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sync' if $m else 'deferred', round(d['ms_per_step'],4), 'filter', round(d['roofline']['avg_launch_ms'],4))"
  done
done
