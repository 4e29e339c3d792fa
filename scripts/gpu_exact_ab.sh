#!/bin/bash
# Exact mode: parity tests, then the 512-slot bench in exact mode (VARIANTS: CWSLG_DEMOD_VARIANT values, 20 = round 1's kernel) and the default bench at 512 slots
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_exact.py tests/test_gpu_e2e_candidates.py tests/test_gpu_demod.py tests/test_gpu_golden.py tests/test_gpu_configs.py tests/test_gpu_lifecycle.py -x -q -m gpu 2>&1 | tail -5
for v in ${VARIANTS:-0 0}; do
  CWSLG_DEMOD_VARIANT=$v timeout 300 python3 bench.py --slots 512 --exact --steps 10 --warmup 3 --no-cpu-baseline --verify 8 > $O/exact_ab_$v.json 2> $O/exact_ab_$v.err || tail -5 $O/exact_ab_$v.err
  python3 - <<PY
import json
d=json.loads(open("$O/exact_ab_$v.json").read().strip().splitlines()[-1])
print("variant $v ms/step %.3f  demod avg_launch %.3f ms frac %.4f  verify %s" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d.get("verify")))
PY
done
for k in 1 2; do
timeout 300 python3 bench.py --slots 512 --steps 20 --warmup 3 --no-cpu-baseline --verify 8 > $O/def512.json 2> $O/def512.err || tail -5 $O/def512.err
python3 - <<PY
import json
d=json.loads(open("$O/def512.json").read().strip().splitlines()[-1])
print("default 512: ms/step %.3f  demod avg_launch %.3f ms frac %.4f whole %.4f verify %s" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"].get("whole_path_frac",0), d.get("verify",{}).get("int16_mismatches")))
PY
done
