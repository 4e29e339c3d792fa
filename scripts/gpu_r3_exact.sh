#!/bin/bash
# Round 3: the new exact-mode kernel (demod_exact3_kernel, product library) against round 2's (lab library, CWSLG_DEMOD_VARIANT=21), same box:
# parity tests first, then 512 slots with and without the sync stage.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_exact.py tests/test_gpu_e2e_candidates.py tests/test_gpu_tune.py tests/test_gpu_lifecycle.py -x -q -m gpu 2>&1 | tail -5
run() { # label, env..., -- bench args
  label=$1; shift
  env "$@" > /dev/null 2>&1
}
for cfg in "exact3|CWSLG_LIB=|--sync 0" "exact2|CWSLG_LIB=lab CWSLG_DEMOD_VARIANT=21|--sync 0" "exact3|CWSLG_LIB=|--sync 0" "exact3+sync|CWSLG_LIB=|--sync 1"; do
  IFS='|' read label envs bargs <<< "$cfg"
  f=$O/r3_exact_${label}.json
  env $envs timeout 300 python3 bench.py --slots 512 --exact --steps 10 --warmup 3 --no-cpu-baseline --verify 8 $bargs > $f 2> $f.err || tail -5 $f.err
  python3 - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1])
r=d["roofline"]
print("%-12s %s ms/step %.3f  demod avg_launch %.3f ms frac %.4f whole %.4f verify %s" % ("$label", r["kernel"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["whole_path_frac"], d.get("verify")))
PY
done
