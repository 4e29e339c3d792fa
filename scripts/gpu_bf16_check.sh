#!/bin/bash
# CWSLG_DEMOD_VARIANT=8 (split-bf16 matrix-core FIR): parity in the default mode, time, package power
cd $GRAFT_REPO_ROOT
CWSLG_DEMOD_VARIANT=8 timeout 900 python -m pytest tests/test_gpu_demod.py tests/test_gpu_golden.py tests/test_gpu_properties.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -6
CWSLG_DEMOD_VARIANT=8 timeout 300 python bench.py --slots 64 --sync 0 --steps 5 --warmup 2 --no-cpu-baseline --verify 8 2>&1 | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('variant 8 verify', d['verify'])"
sample() {
  (env CWSLG_DEMOD_VARIANT=$1 timeout 100 python bench.py --slots 512 --sync 0 --steps 5000 --warmup 2 --no-cpu-baseline --verify 0 > /tmp/pv.json 2>/dev/null) &
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
print('variant $1: demod %.3f ms  power W:$P  sclk MHz:$C' % (r['avg_launch_ms']))"
}
sample 0; sample 8; sample 7
