#!/bin/bash
# quick loop for one kernel change: a parity subset, then bench + kernel stats of the given workloads
#   tools/run_quick.sh <tag> "<pytest -k>" wl1 wl2 ...
tag=$1; expr=$2; shift; shift
out=gpurun_out/r03; mkdir -p $out
if [ -n "$expr" ]; then
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$expr" > $out/${tag}_sel.txt 2>&1; tail -3 $out/${tag}_sel.txt
  grep -q " passed" $out/${tag}_sel.txt && ! grep -q "failed\|error" $out/${tag}_sel.txt || { tail -40 $out/${tag}_sel.txt; exit 1; }
fi
SKIP_TESTS=1 bash tools/run_scaled.sh $tag "$@"
