#!/bin/bash
# Round 5: waves per workgroup of demod_exact5_kernel (-DCWSLG_EXACT5_WAVES: the kernel has no barrier, a workgroup is only the unit of dispatch and of LDS allocation); same box.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-4 2 1 8 4 2 1 8}; do
  export CWSLG_HIPCC_EXTRA="-DCWSLG_EXACT5_WAVES=$v"
  python3 -c "
from cwsl_digi_amd import build as B
B.build(force=True)" > /dev/null 2>&1 || { echo "waves=$v: build failed"; continue; }
  f=$O/r5_x5waves_$v.json
  timeout 300 python3 bench.py --slots 4096 --primary-only --sync 0 --steps 10 --warmup 3 --no-cpu-baseline --verify 4 > $f 2> $f.err || tail -3 $f.err
  python3 - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d["roofline"]
print("EXACT5_WAVES=$v: launch %.3f ms, step %.3f ms, clock %.0f, verify %s" % (r["avg_launch_ms"], d["ms_per_step"], r["valu_pipe"]["clock_mhz"], d["verify"]["int16_mismatches"]))
PY
done
unset CWSLG_HIPCC_EXTRA
