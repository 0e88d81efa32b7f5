#!/bin/bash
# coverage checksum + step time of a few bench runs of one library: bash tools/ck_sum.sh <lib.so> [bench args]
l=$1; shift
for i in 1 2 3; do
  DRPRG_HIP_LIB=$l timeout 300 python bench.py --cpu-sample 0 --e2e 0 --steps 20 --warmup 3 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$l', round(d['ms_per_step'],4), 'dominant', round(d['roofline']['avg_launch_ms'],4), d['config']['coverage_checksum'], d['config']['hits_per_batch'], d['config'].get('full_size_shard_invariance'))"
done
