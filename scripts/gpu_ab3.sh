#!/bin/bash
for W in 2 3 4 5 6 8; do
CWSLG_DEMOD_VARIANT=1 CWSLG_PERSIST_WGS_PER_CU=$W timeout 600 python bench.py --slots 512 --steps 6 --warmup 2 --no-cpu-baseline --sync 0 --verify 0 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); r=j['roofline']
print('persist wgs/cu $W','demod ms %.3f'%r['avg_launch_ms'],'frac %.3f'%r['frac'])
"
done
CWSLG_DEMOD_VARIANT=0 timeout 600 python bench.py --slots 512 --steps 6 --warmup 2 --no-cpu-baseline --sync 0 --verify 0 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); r=j['roofline']
print('tile kernel      ','demod ms %.3f'%r['avg_launch_ms'],'frac %.3f'%r['frac'])
"
