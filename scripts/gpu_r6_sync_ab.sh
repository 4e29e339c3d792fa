#!/bin/bash
# Round 6, sync stage A/B on one box (library rebuilt per variant; VARIANTS = ';'-separated compiler-flag sets, each measured in turn, the list walked
# REPS times).  Fast mode, primary record only: the sync stage is the same in both arithmetic modes.  Then the sync / lifecycle GPU tests on the product build.
O=$GRAFT_REPO_ROOT/gpurun_out/r6ab; mkdir -p $O; cd $GRAFT_REPO_ROOT
IFS=';' read -ra VS <<< "${VARIANTS:--DCWSLG_FUSE_FIN_DEFAULT=0;-DCWSLG_FUSE_FIN_DEFAULT=1}"
for rep in $(seq 1 ${REPS:-2}); do
for v in "${VS[@]}"; do
  export CWSLG_HIPCC_EXTRA="$v"
  python3 -c "
from cwsl_digi_amd import build as B
B.build(force=True)" > /dev/null 2>&1 || { echo "$v: build failed"; continue; }
  f=$O/ab_$(echo "$v" | tr -c 'A-Za-z0-9=\n' '_')_$rep.json
  timeout 300 python3 bench.py --slots 4096 --fast --primary-only --steps 10 --warmup 3 --no-cpu-baseline --verify 0 > $f 2> $f.err || tail -3 $f.err
  python3 - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d["roofline_sync"]; pk=r["per_kernel"]; ro=d["roofline"]
print("%-60s sync %.3f ms (spectra %.3f, search %.3f) + finalise %.3f = %.3f ms; step %.3f ms" % ("$v", r["avg_ms"], pk["spectra"]["avg_ms"], pk["search"]["avg_ms"], ro["finalize_avg_ms"], r["avg_ms"] + ro["finalize_avg_ms"], d["ms_per_step"]))
PY
done; done
unset CWSLG_HIPCC_EXTRA
python3 -c "
from cwsl_digi_amd import build as B
B.build(force=True)" > /dev/null 2>&1
[ "${TESTS:-1}" = 1 ] && timeout 1200 python -m pytest tests/test_gpu_sync.py tests/test_gpu_e2e_candidates.py tests/test_gpu_lifecycle.py tests/test_gpu_configs.py -q -m gpu 2>&1 | tail -4
