#!/bin/bash
# Round 5: demod_exact5_kernel (lane = stream, K = 1 MFMA products) -- parity first (every exact-mode comparison is on bits), then same-box A/B
# against round 4's exact4 (lab library, CWSLG_DEMOD_VARIANT=27) at 512 and 4096 slots, demod only, and the stream length (CWSLG_EXACT5_SEG).
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
if [ "$1" != "bench" ]; then
timeout 1500 python -m pytest tests/test_gpu_exact.py tests/test_gpu_demod.py tests/test_gpu_tune.py tests/test_gpu_adversarial.py tests/test_gpu_properties.py tests/test_gpu_lifecycle.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -15 > $O/r5_exact5_tests.log
tail -15 $O/r5_exact5_tests.log
fi
for cfg in "x5|CWSLG_LIB=|512" "x4|CWSLG_LIB=lab CWSLG_DEMOD_VARIANT=27|512" "x5|CWSLG_LIB=|4096" "x4|CWSLG_LIB=lab CWSLG_DEMOD_VARIANT=27|4096" "x5s704|CWSLG_LIB=lab CWSLG_EXACT5_SEG=704|4096" "x5s2816|CWSLG_LIB=lab CWSLG_EXACT5_SEG=2816|4096" "x5s352|CWSLG_LIB=lab CWSLG_EXACT5_SEG=352|4096"; do
  IFS='|' read label envs slots <<< "$cfg"
  f=$O/r5_exact5_${label}_${slots}.json
  env $envs timeout 300 python3 bench.py --slots $slots --primary-only --sync 0 --steps 10 --warmup 3 --no-cpu-baseline --verify 8 > $f 2> $f.err || tail -5 $f.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$f").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("%-8s %5s slots: %s ms/step %.3f  demod avg_launch %.3f ms frac %.4f clock %s valu_pipe %s verify %s" % ("$label", "$slots", r["kernel"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["valu_pipe"]["clock_mhz"], r["valu_pipe"]["frac"], d.get("verify", {})))
except Exception as e:
    print("$label", "failed", e)
PY
done
