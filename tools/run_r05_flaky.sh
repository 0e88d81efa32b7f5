#!/bin/bash
# round 5's flaky hunt on the final code: tools/flaky_record.sh (cluster-heavy parity tests, every case in both input formats and through the
# opt-in wave form, under DRPRG_HIP_LANES=1..4) and 100 fuzz seeds of the randomized-configuration test, once as they are and once through
# the middle tier of the filter
n=${1:-6}
O=gpurun_out/r05/flaky; mkdir -p $O
bash tools/flaky_record.sh $n $O/flaky_hunt.txt
{
echo "# DRPRG_FUZZ_SEEDS=100 pytest tests/test_gpu_parity.py -k randomized"
DRPRG_FUZZ_SEEDS=100 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k randomized 2>&1 | grep -E " passed| failed| error" | tail -1
echo "# DRPRG_FUZZ_SEEDS=100 DRPRG_FORCE_MID_TIER=1 pytest tests/test_gpu_parity.py -k randomized"
DRPRG_FUZZ_SEEDS=100 DRPRG_FORCE_MID_TIER=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k randomized 2>&1 | grep -E " passed| failed| error" | tail -1
} >> $O/flaky_hunt.txt
tail -8 $O/flaky_hunt.txt
