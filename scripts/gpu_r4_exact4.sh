#!/bin/bash
# Round 4: demod_exact4_kernel (two streams per output pair, four waves per SIMD) -- parity first (every exact-mode comparison is on bits),
# then same-box A/B against round 3's exact3 (lab library, CWSLG_DEMOD_VARIANT=26) at 512 and 4096 slots, demod only and with the sync stage.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_exact.py tests/test_gpu_demod.py tests/test_gpu_tune.py tests/test_gpu_adversarial.py tests/test_gpu_properties.py tests/test_gpu_lifecycle.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -12 > $O/r4_exact4_tests.log
tail -4 $O/r4_exact4_tests.log
for cfg in "x4|CWSLG_LIB=|512" "x3|CWSLG_LIB=lab CWSLG_DEMOD_VARIANT=26|512" "x4|CWSLG_LIB=|4096" "x3|CWSLG_LIB=lab CWSLG_DEMOD_VARIANT=26|4096" "x4|CWSLG_LIB=|4096"; do
  IFS='|' read label envs slots <<< "$cfg"
  f=$O/r4_exact4_${label}_${slots}.json
  env $envs timeout 300 python3 bench.py --slots $slots --primary-only --sync 0 --steps 10 --warmup 3 --no-cpu-baseline --verify 8 > $f 2> $f.err || tail -5 $f.err
  python3 - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1])
r=d["roofline"]
print("%-3s %5s slots: %s ms/step %.3f  demod avg_launch %.3f ms frac %.4f clock %s valu_pipe %s verify %s" % ("$label", "$slots", r["kernel"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["valu_pipe"]["clock_mhz"], r["valu_pipe"]["frac"], d.get("verify", {}).get("max_rel_err")))
PY
done
