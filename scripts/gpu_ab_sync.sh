#!/bin/bash
# A/B of the sync-stage kernels (CWSLG_SYNC_VARIANT bit 0: round-1 Costas search, bit 1: round-1 spectra kernel), same box.
O=$GRAFT_REPO_ROOT/gpurun_out/ab; mkdir -p $O; cd $GRAFT_REPO_ROOT
for S in ${SLOTS_LIST:-512 4096}; do
for V in ${VARIANTS:-0 1 2 3}; do
  CWSLG_SYNC_VARIANT=$V timeout 600 python bench.py --slots $S --steps 10 --warmup 3 --no-cpu-baseline --verify 0 > $O/ab_${S}_v$V.json 2> $O/ab_${S}_v$V.err
  python3 - <<PY
import json
d=json.load(open("$O/ab_${S}_v$V.json")); r=d["roofline"]
print("slots $S variant $V: step %.3f ms  demod %.3f  fin %.3f  sync %.3f  whole %.4f"%(d["ms_per_step"], r["avg_launch_ms"], r["finalize_avg_ms"], r["sync_avg_ms"], r["whole_path_frac"]))
PY
done; done
