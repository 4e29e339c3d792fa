#!/bin/bash
# Round 5: cache policy of finalize_kernel (-DCWSLG_FIN_NT: bit 0 non-temporal loads of the float frame, bit 1 non-temporal stores of the int16 frame); same box.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
for v in 0 1 3 2 0 1 3 2; do
  export CWSLG_HIPCC_EXTRA="-DCWSLG_FIN_NT=$v"
  python3 -c "
from cwsl_digi_amd import build as B
B.build(force=True)" > /dev/null 2>&1 || { echo "fin_nt=$v: build failed"; continue; }
  f=$O/r5_fin_nt_$v.json
  timeout 300 python3 bench.py --slots 4096 --fast --primary-only --steps 10 --warmup 3 --no-cpu-baseline --verify 2 > $f 2> $f.err || tail -3 $f.err
  python3 - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d["roofline"]; s=d["roofline_sync"]
print("FIN_NT=$v: finalise %.4f ms, sync %.3f ms (spectra %.3f), step %.3f ms, verify %s" % (r["finalize_avg_ms"], s["avg_ms"], s["per_kernel"]["spectra"]["avg_ms"], d["ms_per_step"], d["verify"]["max_rel_err"]))
PY
done
unset CWSLG_HIPCC_EXTRA
