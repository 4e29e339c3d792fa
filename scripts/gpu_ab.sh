#!/bin/bash
# tests + A/B of the two demod kernels (sync stage off to isolate the demod kernel)
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for V in 0 1; do
CWSLG_DEMOD_VARIANT=$V timeout 600 python bench.py --slots ${SLOTS:-512} --steps 5 --warmup 2 --no-cpu-baseline --sync 0 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); r=j['roofline']
print('variant $V slots',j['config']['slots_per_gpu'],'Msps %.0f'%j['value'],'ms/step %.3f'%j['ms_per_step'],'demod ms %.3f'%r['avg_launch_ms'],'frac %.3f'%r['frac'],'verify',j['verify'])
"
done
