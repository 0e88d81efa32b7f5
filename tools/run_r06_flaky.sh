#!/bin/bash
# round 6's flaky hunt on the final code.  The chunk schedule makes the ORDER in which sketch_filter_kernel's waves take their tiles depend on
# timing; the result must not.  (1) tools/flaky_record.sh: the cluster-heavy parity tests under DRPRG_HIP_LANES=1..4, n runs each; (2) the whole
# parity file with a chunk schedule forced on every batch large enough for one (two workgroups, min 8 tiles per wave: most of the suite's batches),
# 3 runs, and 100 fuzz seeds the same way; (3) the schedule's own tests 5 times; (4) ten bench runs: the full-size coverage checksum of each.
n=${1:-3}
O=gpurun_out/r06/flaky; mkdir -p $O
bash tools/flaky_record.sh $n $O/flaky_hunt.txt
{
for i in 1 2 3; do
  echo "# run $i: DRPRG_FT_GRID=2 DRPRG_FT_SCHED=128,32,4,8 pytest tests/test_gpu_parity.py -m gpu (every case through a chunk schedule where its batch allows one)"
  DRPRG_FT_GRID=2 DRPRG_FT_SCHED=128,32,4,8 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "not filter_schedule and not extreme_tile and not chunk_schedule" 2>&1 | grep -E " passed| failed| error" | tail -1
done
echo "# DRPRG_FUZZ_SEEDS=100 DRPRG_FT_GRID=3 DRPRG_FT_SCHED=100,20,4,8 pytest tests/test_gpu_parity.py -k randomized"
DRPRG_FUZZ_SEEDS=100 DRPRG_FT_GRID=3 DRPRG_FT_SCHED=100,20,4,8 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k randomized 2>&1 | grep -E " passed| failed| error" | tail -1
for i in 1 2 3 4 5; do
  echo "# run $i: pytest tests/test_gpu_parity.py -k 'filter_schedule or extreme_tile or chunk_schedule'"
  timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "filter_schedule or extreme_tile or chunk_schedule" 2>&1 | grep -E " passed| failed| error" | tail -1
done
echo "# ten bench runs (10 M x 150 bp, 1344+ steps each with the spin-up): ms/step, kernel ms, coverage checksum of the last step, hits per batch"
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 300 python bench.py --cpu-sample 0 --e2e 0 --steps 20 --warmup 5 --no-checks 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['config']['coverage_checksum'], d['config']['hits_per_batch'])"
done
echo "# five bench runs with the packed batch (second stage in the L2): ms/step, kernel ms, coverage checksum of the last step, hits per batch"
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --input packed --cpu-sample 0 --e2e 0 --steps 20 --warmup 5 --no-checks 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['config']['coverage_checksum'], d['config']['hits_per_batch'])"
done
} >> $O/flaky_hunt.txt 2>&1
tail -30 $O/flaky_hunt.txt
