#!/bin/bash
# sketch_filter_kernel's live duration for several DRPRG_FT_SHARE vectors (tiles of the four wave classes of a workgroup)
# usage: tools/ft_share_sweep.sh "<input: ascii|packed>" "<workload>" share1 share2 ...
inp=$1; wl=$2; shift; shift
for sh in "$@"; do
  DRPRG_FT_SHARE=$sh timeout 300 python bench.py --workload $wl --input $inp --steps 60 --warmup 5 --cpu-sample 0 --e2e 0 --no-checks 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$inp $wl share $sh: kernel %.4f ms step %.4f' % (d['roofline']['avg_launch_ms'], d['step_ms']['median']))"
done
