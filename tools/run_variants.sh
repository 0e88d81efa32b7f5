# the GPU parity file under every switch that selects another kernel sequence or host path, then smoke(); failures printed in full
for v in "DRPRG_FT_DEBUG=8" "DRPRG_HIP_LANES=3" "DRPRG_DIRECT_FORM=lds" "DRPRG_WAVE_FUSE=1" "DRPRG_FILTER_FORM=refine" "DRPRG_RC_SLICES=0" "DRPRG_HIP_SPIN=0"; do
  echo "== $v"; env $v timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider > /tmp/rv.log 2>&1; tail -1 /tmp/rv.log
  if grep -q "failed\|error" /tmp/rv.log; then sed -n '/=== FAILURES/,/short test summary/p' /tmp/rv.log | head -120; fi
done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
