# kernel timeline of the last bench steps (rocprofv3 --kernel-trace): bash tools/trace_step.sh [workload] [first kernel of a step]
wl=${1:-mtb}; first=${2:-sketch_filter_kernel}
out=$GRAFT_REPO_ROOT/gpurun_out/trace_$wl; rm -rf $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o $wl -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 4 --warmup 1 --cpu-sample 0 --no-checks --e2e 0 > /dev/null 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
f=$(find $out -name '*kernel_trace.csv' | head -1)
if [ -n "$f" ]; then python tools/step_timeline.py "$f" $first 2 < /dev/null; else echo "no trace written"; fi
