#!/bin/bash
# What the exact-mode FIR loop waits for: stamps of the complete kernel and of diagnostic builds without the scalar tap loads (1),
# without the LDS reads (2), without both (3) inside the loop.  Build here first:  for d in 0 1 2 3: hipcc ... -DCWSLG_STAMP -DCWSLG_X3_DIAG=$d
for d in 0 1 2 3; do
  echo "== CWSLG_X3_DIAG=$d"
  CWSLG_STAMP_LIB=cwsl_digi_amd/lib/libcwslgpu_stamp$d.so timeout 300 python3 scripts/gpu_stamps_exact.py 2>&1 | grep -v amdgpu.ids
done
