#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline --sync 0 --steps 6 --warmup 2 "$@" 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$*','-> demod ms %.3f'%r['avg_launch_ms'],'Gsps %.0f'%(j['value']/1e3),'algo GB/s %.0f'%r['achieved'],'valu TF %.1f'%r['valu_tflops'],j['verify'])
"; }
run --channels-per-rx 1
run --channels-per-rx 8
CWSLG_ITEM_ORDER=1 run --channels-per-rx 8
CWSLG_ITEM_ORDER=2 run --channels-per-rx 1
run --channels-per-rx 8 --slots 4096
