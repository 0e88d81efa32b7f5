#!/bin/bash
# whole GPU suite (round 4)
tag=${1:-t}
O=gpurun_out/r04/$tag; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 > $O/gputests.txt 2>&1; tail -30 $O/gputests.txt
