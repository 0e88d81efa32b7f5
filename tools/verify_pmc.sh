#!/bin/bash
# instruction counts of verify_scan_kernel (SQ counters), whole and with parts switched off: DRPRG_FT_DEBUG 16 no window test, 32 no table
# probe and nothing after it, 512 the slice scan and the candidate positions alone.   usage: tools/verify_pmc.sh wl...
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05/pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in ${@:-mtb}; do for dbg in ${V_DBG:-0 16 32 512}; do
  export DRPRG_FT_DEBUG=$dbg
  rm -rf $O/p; timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $O/p -o p1 -- python3 $R/bench.py --workload $wl --input ${INP:-ascii} --steps 2 --warmup 1 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SMEM --output-format csv -d $O/p -o p2 -- python3 $R/bench.py --workload $wl --input ${INP:-ascii} --steps 2 --warmup 1 --cpu-sample 0 --e2e 0 --no-checks > /dev/null 2>&1
  echo "== $wl debug $dbg"; python3 $R/tools/pmc_summary.py $O/p | grep "verify_scan"
done; done
rm -rf $O/p
