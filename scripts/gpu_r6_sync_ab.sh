#!/bin/bash
# Round 6, sync stage A/B on one box (library rebuilt per variant): finalise fused into the spectra kernel (FUSE) and the rows of passes A / B
# handed to the wave that consumes them (ROWMAP: one workgroup barrier per transform less).  Fast mode, primary record only: the sync stage is the
# same in both arithmetic modes.  Lists stay bit-identical: tests/test_gpu_sync.py on the final variant.
O=$GRAFT_REPO_ROOT/gpurun_out/r6ab; mkdir -p $O; cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-"0 0" "1 0" "0 1" "1 1" "0 0" "1 1"}; do
  set -- $v; F=$1; M=$2
  export CWSLG_HIPCC_EXTRA="-DCWSLG_FUSE_FIN_DEFAULT=$F -DCWSLG_SPEC_ROWMAP=$M ${EXTRA:-}"
  python3 -c "
from cwsl_digi_amd import build as B
B.build(force=True)" > /dev/null 2>&1 || { echo "fuse=$F rowmap=$M: build failed"; continue; }
  f=$O/ab_f${F}_m${M}.json
  timeout 300 python3 bench.py --slots 4096 --fast --primary-only --steps 10 --warmup 3 --no-cpu-baseline --verify 0 > $f 2> $f.err || tail -3 $f.err
  python3 - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d["roofline_sync"]; pk=r["per_kernel"]; ro=d["roofline"]
print("fuse=$F rowmap=$M: sync %.3f ms (spectra %.3f, search %.3f) + finalise %.3f = %.3f ms; step %.3f ms" % (r["avg_ms"], pk["spectra"]["avg_ms"], pk["search"]["avg_ms"], ro["finalize_avg_ms"], r["avg_ms"] + ro["finalize_avg_ms"], d["ms_per_step"]))
PY
done
unset CWSLG_HIPCC_EXTRA
python3 -c "
from cwsl_digi_amd import build as B
B.build(force=True)" > /dev/null 2>&1
timeout 1200 python -m pytest tests/test_gpu_sync.py tests/test_gpu_e2e_candidates.py tests/test_gpu_lifecycle.py tests/test_gpu_configs.py -q -m gpu 2>&1 | tail -4
