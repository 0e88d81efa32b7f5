DRPRG_WAVE_DEBUG=4 python bench.py --workload big --steps 2 --warmup 1 --cpu-sample 0 --no-checks 2>&1 | grep -v "^{" | tail -5
