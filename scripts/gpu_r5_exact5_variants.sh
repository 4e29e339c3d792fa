#!/bin/bash
# Round 5: same-box A/B of demod_exact5_kernel's generator switches (the library is rebuilt on the box for each): 4096 slots, demod only.
# usage: gpu_r5_exact5_variants.sh "LABEL|ENV=.. ENV=.." ...
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
cp cwsl_digi_amd/csrc/exact5_asm.inc /tmp/x5_keep.inc
for cfg in "$@" "base|"; do
  IFS='|' read label envs <<< "$cfg"
  env $envs python3 scripts/gen_exact5_asm.py > cwsl_digi_amd/csrc/exact5_asm.inc
  [ "$label" = "base" ] && cp /tmp/x5_keep.inc cwsl_digi_amd/csrc/exact5_asm.inc
  python3 -c "
from cwsl_digi_amd import build as B
B.build(force=True)" > /dev/null 2>&1 || { echo "$label: build failed"; continue; }
  for rep in 1 2; do
  f=$O/r5_x5var_${label}.json
  timeout 300 python3 bench.py --slots 4096 --primary-only --sync 0 --steps 10 --warmup 3 --no-cpu-baseline --verify 4 > $f 2> $f.err || tail -5 $f.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$f").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("%-12s %s ms/step %.3f clock %.0f verify %s" % ("$label", r["kernel"], d["ms_per_step"], r["valu_pipe"]["clock_mhz"], d.get("verify", {}).get("int16_mismatches")))
except Exception as e:
    print("$label", "failed", e)
PY
  done
done
