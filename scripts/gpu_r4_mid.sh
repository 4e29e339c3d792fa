#!/bin/bash
# Round 4, mid-round pass: soak with the new exact kernel, the reference topology (32 receivers x 128 FT8 channels) through bench.py,
# and BASELINE configs[2] / configs[4] at full size.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 500 python3 scripts/gpu_soak.py --seconds 360 --seed 11 > $O/r4_soak.json 2> $O/r4_soak.err; tail -c 900 $O/r4_soak.json; tail -3 $O/r4_soak.err
timeout 600 python3 bench.py --channels-per-rx 128 --steps 5 --warmup 2 --no-cpu-baseline > $O/r4_shared_32x128.json 2> $O/r4_shared_32x128.err || tail -5 $O/r4_shared_32x128.err
python3 - <<PY
import json
d=json.loads(open("$O/r4_shared_32x128.json").read().strip().splitlines()[-1])
print("shared 32x128 exact:", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["shared_topology"], d["verify"])
f=d.get("fast")
if f: print("shared 32x128 fast:", f["value"], f["ms_per_step"], f["roofline"]["avg_launch_ms"], f["verify"])
PY
timeout 900 python3 scripts/run_configs.py --config 3 --steps 3 > $O/r4_config3.json 2> $O/r4_config3.err; tail -c 1200 $O/r4_config3.json; tail -2 $O/r4_config3.err
timeout 900 python3 scripts/run_configs.py --config 5 --steps 2 > $O/r4_config5.json 2> $O/r4_config5.err; tail -c 1200 $O/r4_config5.json; tail -2 $O/r4_config5.err
