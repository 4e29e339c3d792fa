#!/bin/bash
# Where the power goes: the split-bf16 kernel (CWSLG_DEMOD_VARIANT=8) complete, without its matrix instructions (diag1), and also
# without the split arithmetic (diag2), against the traffic-only probe (variant 9).  512 slots, demod only; time, power, clock.
cd $GRAFT_REPO_ROOT
sample() { # label variant lib
  (env CWSLG_DEMOD_VARIANT=$2 CWSLG_LIB=$3 timeout 100 python bench.py --slots 512 --sync 0 --steps 5000 --warmup 2 --no-cpu-baseline --verify 0 > /tmp/pv.json 2>/dev/null) &
  BP=$!
  sleep 9
  P=""; C=""
  for k in 1 2 3; do
    L=$(rocm-smi --showpower --showclocks 2>/dev/null | tr '\n' ';')
    P="$P $(echo "$L" | sed -n 's/.*Package Power (W): \([0-9.]*\).*/\1/p')"
    C="$C $(echo "$L" | sed -n 's/.*sclk clock level: [^(]*(\([0-9]*\)Mhz).*/\1/p')"
    sleep 1.5
  done
  wait $BP
  python3 -c "
import json; d=json.load(open('/tmp/pv.json')); r=d['roofline']
print('%-44s %.3f ms  power W:$P  sclk MHz:$C' % ('$1', r['avg_launch_ms']))"
}
A=$GRAFT_REPO_ROOT/cwsl_digi_amd/lib/ab
sample "probe: loads + stores only (9)" 9 ""
sample "bf16 kernel, no MFMA, no split (diag2)" 8 $A/diag2.so
sample "bf16 kernel, no MFMA (diag1)" 8 $A/diag1.so
sample "bf16 kernel complete (8)" 8 ""
sample "demod_kernel (0)" 0 ""
