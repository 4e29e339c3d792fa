#!/bin/bash
# Round 5: cache policy of the FT8 spectra plane (written once by symbol_spectra_v2_kernel, read once by ft8_sync_chan_kernel): same-box A/B of
# -DCWSLG_PLANE_NT=0..3 (bit 0 non-temporal stores, bit 1 non-temporal loads); the library is rebuilt on the box for each.  Lists stay bit-identical
# (tests/test_gpu_sync.py first).
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
for v in 0 3 1 2 0 3; do
  export CWSLG_HIPCC_EXTRA="-DCWSLG_PLANE_NT=$v"
  python3 -c "
from cwsl_digi_amd import build as B
B.build(force=True)" > /dev/null 2>&1 || { echo "nt=$v: build failed"; continue; }
  [ $v = 3 ] && timeout 600 python -m pytest tests/test_gpu_sync.py -x -q -m gpu 2>&1 | tail -1
  f=$O/r5_plane_nt_$v.json
  timeout 300 python3 bench.py --slots 4096 --fast --primary-only --steps 10 --warmup 3 --no-cpu-baseline --verify 0 > $f 2> $f.err || tail -3 $f.err
  python3 - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d["roofline_sync"]; pk=r["per_kernel"]
print("PLANE_NT=$v: sync %.3f ms (spectra %.3f, search %.3f), step %.3f ms" % (r["avg_ms"], pk["spectra"]["avg_ms"], pk["search"]["avg_ms"], d["ms_per_step"]))
PY
done
unset CWSLG_HIPCC_EXTRA
