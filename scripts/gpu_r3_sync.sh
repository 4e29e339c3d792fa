#!/bin/bash
# Round 3 sync stage: the fused per-channel Costas search + candidate selection (product) against round 2's three-launch form (lab
# library, CWSLG_SYNC_VARIANT=64), same box: parity tests first, then the default bench at 512 and 4096 slots (fast mode only).
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sync.py tests/test_gpu_e2e_candidates.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -3
for cfg in "product|CWSLG_LIB=|512" "chanfused|CWSLG_LIB=lab CWSLG_SYNC_VARIANT=128|512" "product|CWSLG_LIB=|4096" "chanfused|CWSLG_LIB=lab CWSLG_SYNC_VARIANT=128|4096" "product|CWSLG_LIB=|4096"; do
  IFS='|' read label envs slots <<< "$cfg"
  f=$O/r3_sync_${label}_${slots}.json
  env $envs timeout 300 python3 bench.py --slots $slots --fast-only --steps 10 --warmup 3 --no-cpu-baseline --verify 0 > $f 2> $f.err || tail -5 $f.err
  python3 - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1])
r=d["roofline"]
print("%-10s %5s slots: ms/step %.3f  demod %.3f  finalize %.3f  sync %.3f  whole %.4f" % ("$label", "$slots", d["ms_per_step"], r["avg_launch_ms"], r["finalize_avg_ms"], r["sync_avg_ms"], r["whole_path_frac"]))
PY
done
