#!/bin/bash
# Round 5: outputs per stream (seg_len) of demod_exact5_kernel, lab library (CWSLG_EXACT5_SEG caps it), 4096 slots, demod only, same box.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
for seg in ${@:-1408 704 2816 5632 1408}; do
  for rep in 1 2; do
  f=$O/r5_seg_$seg.json
  CWSLG_LIB=lab CWSLG_EXACT5_SEG=$seg timeout 300 python3 bench.py --slots 4096 --primary-only --sync 0 --steps 10 --warmup 3 --no-cpu-baseline --verify 4 > $f 2> $f.err || tail -5 $f.err
  python3 - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1])
r=d["roofline"]
print("seg %-6s %s launch %.3f ms step %.3f clock %.0f verify %s" % ("$seg", r["kernel"], r["avg_launch_ms"], d["ms_per_step"], r["valu_pipe"]["clock_mhz"], d.get("verify", {}).get("int16_mismatches")))
PY
  done
done
