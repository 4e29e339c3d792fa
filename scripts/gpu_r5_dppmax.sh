#!/bin/bash
# Round 5: the search's two per-bin reductions as one interleaved chain of fused DPP maxima (wave_first_max2, -DCWSLG_SEARCH_DPPMAX=1, the product) against
# hipcc's wave_first_max twice (=0): same box, library rebuilt per variant; parity of the product form first.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sync.py tests/test_gpu_e2e_candidates.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -1
for v in ${VARIANTS:-0 1 0 1}; do
  export CWSLG_HIPCC_EXTRA="-D${SWITCH:-CWSLG_SEARCH_DPPMAX}=$v"
  python3 -c "
from cwsl_digi_amd import build as B
B.build(force=True)" > /dev/null 2>&1 || { echo "dppmax=$v: build failed"; continue; }
  f=$O/r5_dppmax_$v.json
  timeout 300 python3 bench.py --slots 4096 --fast --primary-only --steps 10 --warmup 3 --no-cpu-baseline --verify 0 > $f 2> $f.err || tail -3 $f.err
  python3 - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d["roofline_sync"]; pk=r["per_kernel"]
print("${SWITCH:-CWSLG_SEARCH_DPPMAX}=$v: sync %.3f ms (spectra %.3f, search %.3f), step %.3f ms" % (r["avg_ms"], pk["spectra"]["avg_ms"], pk["search"]["avg_ms"], d["ms_per_step"]))
PY
done
unset CWSLG_HIPCC_EXTRA
