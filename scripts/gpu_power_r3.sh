#!/bin/bash
# Round 3: package power and shader clock (rocm-smi) while ONE kernel form loops: exact mode without the sync stage (demod_exact3_kernel,
# the clock bench.py's valu_pipe record assumes), fast mode without it (demod_kernel), and the whole default step in both modes.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
sample() { # label, bench args...
  label=$1; shift
  timeout 150 python bench.py --warmup 2 --no-cpu-baseline --verify 0 "$@" > $O/power_$label.json 2>/dev/null &
  BP=$!
  sleep 30
  echo "== $label"
  for k in 1 2 3 4 5; do
    rocm-smi --showpower --showclocks 2>&1 | grep -i "power (W)\|sclk" | sed 's/^.*: //' | tr '\n' ';'; echo
    sleep 2
  done
  wait $BP
  python3 -c "
import json,sys
d=json.loads(open('$O/power_$label.json').read().strip().splitlines()[-1]); r=d['roofline']
print('   %s: %.3f ms per step, demod %.3f ms per launch' % (r['kernel'], d['ms_per_step'], r['avg_launch_ms']))"
}
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -i "power\|sclk" | head -6
sample exact_demod_only --exact --sync 0 --slots 512 --steps 9000
sample fast_demod_only --fast-only --sync 0 --slots 512 --steps 16000
sample exact_step --exact --slots 4096 --steps 1000
sample fast_step --fast-only --slots 4096 --steps 1500
