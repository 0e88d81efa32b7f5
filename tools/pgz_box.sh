#!/bin/bash
# parallel-gunzip timing on the GPU box's host cores: 4M-read FASTQ, gzip -6, tools/pgz_bench.cpp for several thread counts / chunk sizes
set -e
cd "$(dirname "$0")/.."
g++ -O2 -std=c++17 -I drprg_amd/csrc -o /tmp/pgz_bench tools/pgz_bench.cpp drprg_amd/csrc/pgunzip.cpp -lz -ldl -lpthread
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from drprg_amd import synth
rng = np.random.default_rng(3)
n = 4_000_000
bases = np.frombuffer(b'ACGT', np.uint8)[rng.integers(0, 4, n * 150)]
synth.write_fastq_fixed('/dev/shm/pgz.fq', bases, 150)
PY
gzip -6 -c /dev/shm/pgz.fq > /dev/shm/pgz.fq.gz
ls -l /dev/shm/pgz.fq /dev/shm/pgz.fq.gz
nproc
( time gzip -dc /dev/shm/pgz.fq.gz > /dev/null ) 2>&1 | grep real
for t in 1 4 8 16 32 64; do DRPRG_GZ_DEBUG=1 /tmp/pgz_bench /dev/shm/pgz.fq.gz $t 2>&1 | tail -2; done
for c in 262144 524288 1048576 4194304; do echo chunk $c; DRPRG_GZ_DEBUG=1 /tmp/pgz_bench /dev/shm/pgz.fq.gz 32 $c 2>&1 | tail -2; done
rm -f /dev/shm/pgz.fq /dev/shm/pgz.fq.gz
