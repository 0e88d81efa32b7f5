#!/bin/bash
# selected GPU tests (pytest -k expression) under a timeout: tools/run_sel.sh <tag> "<-k expr>" [file]
tag=$1; expr=$2; file=${3:-tests}
out=gpurun_out/r03; mkdir -p $out
timeout 900 python -m pytest $file -x -q -m gpu -k "$expr" > $out/${tag}_sel.txt 2>&1; tail -25 $out/${tag}_sel.txt
