#!/bin/bash
# Round 5: symbol_spectra_v2_kernel at four (126 VGPRs, no scratch) against five waves per SIMD (96 VGPRs, 92 bytes of scratch per lane): -DCWSLG_SPEC_WAVES,
# same box, library rebuilt per variant; lists stay bit-identical (tests/test_gpu_sync.py).
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
for v in 4 5 4 5; do
  export CWSLG_HIPCC_EXTRA="-DCWSLG_SPEC_WAVES=$v"
  python3 -c "
from cwsl_digi_amd import build as B
B.build(force=True)" > /dev/null 2>&1 || { echo "waves=$v: build failed"; continue; }
  [ $v = 5 ] && timeout 600 python -m pytest tests/test_gpu_sync.py -x -q -m gpu 2>&1 | tail -1
  f=$O/r5_spec_waves_$v.json
  timeout 300 python3 bench.py --slots 4096 --fast --primary-only --steps 10 --warmup 3 --no-cpu-baseline --verify 0 > $f 2> $f.err || tail -3 $f.err
  python3 - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d["roofline_sync"]; pk=r["per_kernel"]
print("SPEC_WAVES=$v: sync %.3f ms (spectra %.3f, search %.3f), step %.3f ms" % (r["avg_ms"], pk["spectra"]["avg_ms"], pk["search"]["avg_ms"], d["ms_per_step"]))
PY
done
unset CWSLG_HIPCC_EXTRA
