#!/bin/bash
# Round 3, exact mode at 192 kHz: the FIR as one generated assembly statement with fixed registers and 16-byte LDS reads (product) against
# the C++ form with hand-issued 8-byte reads (lab library, CWSLG_DEMOD_VARIANT=25), same box: parity tests first (exact mode: bit-identical
# frames), then 512 and 4096 slots without the sync stage, then the default bench line.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_exact.py tests/test_gpu_demod.py tests/test_gpu_tune.py tests/test_gpu_adversarial.py tests/test_gpu_properties.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
for cfg in "asm|CWSLG_LIB=|512" "c++|CWSLG_LIB=lab CWSLG_DEMOD_VARIANT=25|512" "asm|CWSLG_LIB=|4096" "c++|CWSLG_LIB=lab CWSLG_DEMOD_VARIANT=25|4096" "asm|CWSLG_LIB=|4096"; do
  IFS='|' read label envs slots <<< "$cfg"
  f=$O/r3_exactasm_${label}_${slots}.json
  env $envs timeout 300 python3 bench.py --slots $slots --exact --sync 0 --steps 10 --warmup 3 --no-cpu-baseline --verify 8 > $f 2> $f.err || tail -5 $f.err
  python3 - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1])
r=d["roofline"]
print("%-5s %5s slots: %s ms/step %.3f  demod avg_launch %.3f ms frac %.4f verify %s" % ("$label", "$slots", r["kernel"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], d.get("verify", {}).get("max_rel_err")))
PY
done
