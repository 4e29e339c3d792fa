#!/bin/bash
# Round 5: non-temporal loads of the IQ ring in demod_kernel (fast mode): same-box A/B of -DCWSLG_RING_NT=0/1, 4096 slots, library rebuilt per variant.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  export CWSLG_HIPCC_EXTRA="-DCWSLG_RING_NT=$v"
  python3 -c "
from cwsl_digi_amd import build as B
B.build(force=True)" > /dev/null 2>&1 || { echo "nt=$v: build failed"; continue; }
  f=$O/r5_fast_nt_$v.json
  timeout 300 python3 bench.py --slots 4096 --fast --primary-only --steps 10 --warmup 3 --no-cpu-baseline --verify 4 > $f 2> $f.err || tail -3 $f.err
  python3 - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d["roofline"]
print("RING_NT=$v: demod %.3f ms (frac %.4f) step %.3f ms whole path %.4f clock %s verify %s" % (r["avg_launch_ms"], r["frac"], d["ms_per_step"], r["whole_path_frac"], r.get("clock_mhz"), d["verify"]["max_rel_err"]))
PY
done
unset CWSLG_HIPCC_EXTRA
