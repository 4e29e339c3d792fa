#!/bin/bash
# Round 3 Costas search, same box: ft8_sync_chan_kernel (per-channel sliding window + hand-scheduled LDS stream + fused candidate
# selection: the product), round 2's three launches (lab library, CWSLG_SYNC_VARIANT=64: ft8_sync2d_v2_kernel) and the hand-scheduled
# search as its own launch (lab, CWSLG_SYNC_VARIANT=128: ft8_sync2d_v3_kernel).  Parity tests first (product, then both lab forms),
# then the bench at 512 and 4096 slots (fast mode only) and the per-kernel averages of one profiled run of each.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; R=$GRAFT_REPO_ROOT
T="tests/test_gpu_sync.py tests/test_gpu_e2e_candidates.py tests/test_gpu_fullsize.py"
timeout 1200 python -m pytest $T -x -q -m gpu 2>&1 | grep -v "^$" | tail -3
CWSLG_LIB=lab CWSLG_SYNC_VARIANT=128 timeout 1200 python -m pytest $T -x -q -m gpu 2>&1 | grep -v "^$" | tail -3
CWSLG_LIB=lab CWSLG_SYNC_VARIANT=64 timeout 1200 python -m pytest $T -x -q -m gpu 2>&1 | grep -v "^$" | tail -3
for cfg in "product|CWSLG_LIB=|512" "round2|CWSLG_LIB=lab CWSLG_SYNC_VARIANT=64|512" "v3|CWSLG_LIB=lab CWSLG_SYNC_VARIANT=128|512" "product|CWSLG_LIB=|4096" "round2|CWSLG_LIB=lab CWSLG_SYNC_VARIANT=64|4096" "v3|CWSLG_LIB=lab CWSLG_SYNC_VARIANT=128|4096" "product|CWSLG_LIB=|4096"; do
  IFS='|' read label envs slots <<< "$cfg"
  f=$O/r3_sync2d_${label}_${slots}.json
  env $envs timeout 300 python3 bench.py --slots $slots --fast-only --steps 10 --warmup 3 --no-cpu-baseline --verify 0 > $f 2> $f.err || tail -5 $f.err
  python3 - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1])
r=d["roofline"]
print("%-10s %5s slots: ms/step %.3f  demod %.3f  finalize %.3f  sync %.3f  whole %.4f" % ("$label", "$slots", d["ms_per_step"], r["avg_launch_ms"], r["finalize_avg_ms"], r["sync_avg_ms"], r["whole_path_frac"]))
PY
done
export TMPDIR=/tmp; cd /tmp
for v in product round2 v3; do
  unset CWSLG_LIB CWSLG_SYNC_VARIANT
  if [ $v = round2 ]; then export CWSLG_LIB=lab CWSLG_SYNC_VARIANT=64; fi
  if [ $v = v3 ]; then export CWSLG_LIB=lab CWSLG_SYNC_VARIANT=128; fi
  rm -rf $O/prof_$v
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 $R/bench.py --no-cpu-baseline --verify 0 --steps 3 --warmup 1 --fast-only > $O/prof_$v.log 2>&1
  f=$(find $O/prof_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('spectra','sync','candidates','demod_kernel','finalize')):
        print("  %-72s %3s calls  %.3f ms" % (n[:72], r['Calls'], float(r['AverageNs'])/1e6))
PY
done
