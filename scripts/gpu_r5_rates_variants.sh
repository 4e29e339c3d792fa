#!/bin/bash
# Round 5: the generator switches of demod_exact5_kernel at 48 / 96 / 192 kHz (512 slots, exact mode), library rebuilt on the box per variant.
# usage: gpu_r5_rates_variants.sh "LABEL|ENV=.. ENV=.." ...
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
cp cwsl_digi_amd/csrc/exact5_asm.inc /tmp/x5_keep.inc
for cfg in "$@" "base|" "$@" "base|"; do
  IFS='|' read label envs <<< "$cfg"
  env $envs python3 scripts/gen_exact5_asm.py > cwsl_digi_amd/csrc/exact5_asm.inc
  [ "$label" = "base" ] && cp /tmp/x5_keep.inc cwsl_digi_amd/csrc/exact5_asm.inc
  python3 -c "
from cwsl_digi_amd import build as B
B.build(force=True)" > /dev/null 2>&1 || { echo "$label: build failed"; continue; }
  echo "== $label ($envs)"
  timeout 600 python3 scripts/gpu_rates_exact.py 2>&1 | grep "^fs" | grep exact
done
cp /tmp/x5_keep.inc cwsl_digi_amd/csrc/exact5_asm.inc
