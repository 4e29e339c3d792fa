#!/bin/bash
# Round 5: demod launch time at 48 / 96 / 192 kHz, exact mode: demod_exact5_kernel<D> (product) against round 3/4's kernels (lab, CWSLG_DEMOD_VARIANT=26:
# demod_exact3_kernel at every rate), and the fast mode for scale.  512 FT8 slots, no sync stage.  scripts/gpu_rates_exact.py.
O=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
{ echo "== product library"; timeout 900 python scripts/gpu_rates_exact.py 2>&1 | grep "^fs"
  echo "== lab library, CWSLG_DEMOD_VARIANT=26 (demod_exact3_kernel at every rate)"; CWSLG_LIB=lab CWSLG_DEMOD_VARIANT=26 timeout 900 python scripts/gpu_rates_exact.py 2>&1 | grep "^fs" | grep exact
} > $O/r5_rates.txt 2>&1
cat $O/r5_rates.txt
