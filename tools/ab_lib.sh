# same-box A/B of two builds of the library: bash tools/ab_lib.sh build/lib_old.so build/lib_new.so [workload]
wl=${3:-mtb}
for i in 1 2 3; do
  for l in "$1" "$2"; do
    DRPRG_HIP_LIB=$l timeout 300 python bench.py --workload $wl --cpu-sample 0 --e2e 0 --no-checks --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$l', round(d['ms_per_step'],4), 'dominant', round(d['roofline']['avg_launch_ms'],4), d['config']['coverage_checksum'])"
  done
done
