# one test many times: bash tools/flaky_one.sh <pytest -k expression> <runs> [ENV=..]
k="$1"; n="$2"; shift 2
fail=0
for i in $(seq 1 $n); do
  env "$@" timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "$k" > /tmp/fo.log 2>&1
  if ! grep -q " passed" /tmp/fo.log || grep -q "failed" /tmp/fo.log; then fail=$((fail+1)); echo "run $i FAILED"; sed -n '/=== FAILURES/,/short test summary/p' /tmp/fo.log | head -60; fi
done
echo "$k: $fail failures in $n runs ($*)"
