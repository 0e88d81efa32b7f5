# runs the GPU parity file repeatedly (optionally under env settings given as arguments) and prints any failure in full
n=${N:-6}
for i in $(seq 1 $n); do
  env "$@" timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider 2>&1 > /tmp/fh_$i.log
  tail -1 /tmp/fh_$i.log
  if grep -q "failed" /tmp/fh_$i.log; then grep -n "^FAILED\|Error\|assert" /tmp/fh_$i.log | head -20; sed -n '/=== FAILURES/,/short test summary/p' /tmp/fh_$i.log | head -80; fi
done
