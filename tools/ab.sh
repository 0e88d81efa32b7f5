python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for d in 0 0 0; do DRPRG_FT_DEBUG=$d python bench.py --cpu-sample 0 2>&1 | grep metric | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('debug=$d', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['config']['full_size_direct_vs_filtered_kernel_identical'])"; done
