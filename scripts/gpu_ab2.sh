#!/bin/bash
# interleaved A/B of two library builds on the same box (sync off: demod kernel only)
for rep in 1 2 3; do
for L in libcwslgpu_prev.so libcwslgpu.so; do
CWSLG_LIB=$GRAFT_REPO_ROOT/cwsl_digi_amd/lib/$L timeout 600 python bench.py --slots ${SLOTS:-512} --steps 8 --warmup 2 --no-cpu-baseline --sync 0 --verify 0 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$L','Msps %.0f'%j['value'],'ms/step %.3f'%j['ms_per_step'],'demod ms %.3f'%r['avg_launch_ms'],'frac %.3f'%r['frac'])
"
done; done
