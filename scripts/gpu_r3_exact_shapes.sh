#!/bin/bash
# demod_exact3_kernel: workgroup shapes, same box (lab library): 2 waves x 256 outputs (default), 4 waves x 512 (variant 23), 1 wave x 128 (24)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
for v in ${VARIANTS:-0 23 24 0 23 24}; do
  f=$O/r3_shape_$v.json
  CWSLG_LIB=lab CWSLG_DEMOD_VARIANT=$v timeout 300 python3 bench.py --slots ${S:-512} --exact --steps 10 --warmup 3 --no-cpu-baseline --verify 4 --sync 0 > $f 2> $f.err || tail -5 $f.err
  python3 - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1])
r=d["roofline"]
print("variant %-3s ms/step %.3f  demod avg_launch %.3f ms frac %.4f verify %s" % ("$v", d["ms_per_step"], r["avg_launch_ms"], r["frac"], d.get("verify")))
PY
done
