#!/bin/bash
# Round 5: the search stream's repeated address addition hoisted (X2D_HOIST=1, product: 25 instead of 31 v_add_u32 per bin) against round 3's form (=0); same box,
# sync2d_asm.inc regenerated and the library rebuilt per variant.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
cp cwsl_digi_amd/csrc/sync2d_asm.inc /tmp/x2d_keep.inc
for v in 0 1 0 1 0 1; do
  X2D_HOIST=$v python3 scripts/gen_sync2d_asm.py > cwsl_digi_amd/csrc/sync2d_asm.inc
  python3 -c "
from cwsl_digi_amd import build as B
B.build(force=True)" > /dev/null 2>&1 || { echo "hoist=$v: build failed"; continue; }
  f=$O/r5_x2d_$v.json
  timeout 300 python3 bench.py --slots 4096 --fast --primary-only --steps 10 --warmup 3 --no-cpu-baseline --verify 0 > $f 2> $f.err || tail -3 $f.err
  python3 - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d["roofline_sync"]; pk=r["per_kernel"]
print("X2D_HOIST=$v: sync %.3f ms (spectra %.3f, search %.3f), step %.3f ms" % (r["avg_ms"], pk["spectra"]["avg_ms"], pk["search"]["avg_ms"], d["ms_per_step"]))
PY
done
cp /tmp/x2d_keep.inc cwsl_digi_amd/csrc/sync2d_asm.inc
