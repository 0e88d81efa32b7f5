#!/bin/bash
# one bench run of the default library: step, dominant kernel, the filter's class end times: bash tools/ck_sched.sh [bench args]   (env decides the variant)
timeout 300 python bench.py --cpu-sample 0 --e2e 0 --steps 20 --warmup 3 --no-checks "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; f=r.get('filter_schedule') or {}
print(round(d['ms_per_step'],4), 'dominant', round(r['avg_launch_ms'],4), d['config']['coverage_checksum'], 'form', f.get('form'), 'shares', f.get('round0_class_shares_per_256'), 'chunks', f.get('chunk_tiles_per_round'), 'ends', f.get('class_end_us'))"
