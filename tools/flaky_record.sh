#!/bin/bash
# profiles/<round>/flaky_hunt.txt: consecutive runs of the GPU parity file's cluster-heavy tests under DRPRG_HIP_LANES=1..4 (read ranges on
# concurrent streams: the setting under which read_cluster_kernel's chunks of neighbouring ranges run side by side)
# usage: bash tools/flaky_record.sh <runs per lane setting> <out file>
n=${1:-13}; out=${2:-gpurun_out/r03/flaky_hunt.txt}
mkdir -p $(dirname $out)
k="config1 or config2 or several_groups or randomized or dense or long_reads or short_reads or do_not_fit or staged_range or many_small or batches or deferred or second_stage or middle_tier or scaled or multi_device"
echo "# $(date -u +%FT%TZ)  pytest tests/test_gpu_parity.py -m gpu -k \"$k\"" > $out
total=0; bad=0
for lanes in 1 2 3 4; do
  for i in $(seq 1 $n); do
    DRPRG_HIP_LANES=$lanes DRPRG_HIP_LANES_MIN_BASES=0 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "$k" > /tmp/fr.log 2>&1
    line=$(grep -E " passed| failed| error" /tmp/fr.log | tail -1)
    total=$((total+1))
    if ! echo "$line" | grep -q " passed" || echo "$line" | grep -q "failed\|error"; then bad=$((bad+1)); sed -n '/=== FAILURES/,/short test summary/p' /tmp/fr.log | head -60 >> $out; fi
    echo "DRPRG_HIP_LANES=$lanes run $i: $line" >> $out
  done
done
echo "# $total runs, $bad not green" >> $out
# ... and the cold-context stress of the multi-device ingest (tools/stress_multi.py: fresh contexts every round, device 0 listed three / two times / once)
echo "# STRESS_COLD=1 python tools/stress_multi.py 100" >> $out
STRESS_COLD=1 timeout 600 python tools/stress_multi.py 100 2>&1 | grep -v amdgpu.ids >> $out
tail -3 $out
