#!/bin/bash
# Round 5: the whole library with hipcc's SLP vectoriser on (-fslp-vectorize: packed FP32 where it finds pairs) against the product's -fno-slp-vectorize, same box:
# does fewer, packed instructions help the sync stage (as they did the exact demodulator)?  Lists must stay bit-identical (tests/test_gpu_sync.py).
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
for v in off on off on; do
  if [ $v = on ]; then export CWSLG_HIPCC_EXTRA="-fslp-vectorize"; else unset CWSLG_HIPCC_EXTRA; fi
  python3 -c "
from cwsl_digi_amd import build as B
B.build(force=True)" > /dev/null 2>&1 || { echo "slp=$v: build failed"; continue; }
  [ $v = on ] && timeout 600 python -m pytest tests/test_gpu_sync.py -x -q -m gpu 2>&1 | tail -1
  f=$O/r5_slp_$v.json
  timeout 300 python3 bench.py --slots 4096 --fast --primary-only --steps 10 --warmup 3 --no-cpu-baseline --verify 0 > $f 2> $f.err || tail -3 $f.err
  python3 - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d["roofline_sync"]; pk=r["per_kernel"]
print("slp=$v: sync %.3f ms (spectra %.3f, search %.3f), fast demod %.3f ms, step %.3f ms" % (r["avg_ms"], pk["spectra"]["avg_ms"], pk["search"]["avg_ms"], d["roofline"]["avg_launch_ms"], d["ms_per_step"]))
PY
done
unset CWSLG_HIPCC_EXTRA
