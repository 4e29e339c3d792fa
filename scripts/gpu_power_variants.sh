#!/bin/bash
# Package power and shader clock per kernel form: 512 slots, demod only (--sync 0), sampled while the bench loops.
#   9 = traffic-only probe (same loads and stores, no arithmetic), 0 = demod_kernel, 7 = FIR on the matrix cores, exact = exact mode
cd $GRAFT_REPO_ROOT
sample() { # label, env, args
  (env $2 timeout 100 python bench.py --slots 512 --sync 0 --steps 6000 --warmup 2 --no-cpu-baseline --verify 0 $3 > /tmp/pv.json 2>/dev/null) &
  BP=$!
  sleep 9
  P=""; C=""
  for k in 1 2 3 4; do
    L=$(rocm-smi --showpower --showclocks 2>/dev/null | tr '\n' ';')
    P="$P $(echo "$L" | sed -n 's/.*Package Power (W): \([0-9.]*\).*/\1/p')"
    C="$C $(echo "$L" | sed -n 's/.*sclk clock level: [^(]*(\([0-9]*\)Mhz).*/\1/p')"
    sleep 1.5
  done
  wait $BP
  python3 -c "
import json; d=json.load(open('/tmp/pv.json')); r=d['roofline']
print('%-22s demod %.3f ms  power W:$P  sclk MHz:$C' % ('$1', r['avg_launch_ms']))"
}
sample "probe (variant 9)" CWSLG_DEMOD_VARIANT=9 ""
sample "demod_kernel" CWSLG_DEMOD_VARIANT=0 ""
sample "MFMA FIR (variant 7)" CWSLG_DEMOD_VARIANT=7 ""
sample "exact mode" CWSLG_DEMOD_VARIANT=0 "--exact"
