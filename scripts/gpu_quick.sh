#!/bin/bash
# quick GPU iteration: parity tests + bench at 512 slots (no CPU baseline)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for S in ${SLOTS:-512}; do
timeout 600 python bench.py --slots $S --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); r=j['roofline']
print('slots',j['config']['slots_per_gpu'],'Msps %.0f'%j['value'],'ms/step %.3f'%j['ms_per_step'],'demod ms %.3f'%r['avg_launch_ms'],'frac %.3f'%r['frac'],'fin ms %.3f'%r['finalize_avg_ms'],'sync ms %.3f'%r.get('sync_avg_ms',0),'verify',j['verify'])
"
done
