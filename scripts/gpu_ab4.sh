#!/bin/bash
# A/B of demod variants on the same box: 0 tile, 2 persistent loop (several residency settings), interleaved
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --sync 0 --verify 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 demod %.3f ms frac %.3f err %.2e' % (r['avg_launch_ms'], r['frac'], d['verify']['max_rel_err']))"; }
for rep in 1 2; do
CWSLG_DEMOD_VARIANT=0 run v0
CWSLG_DEMOD_VARIANT=2 run v2-auto
CWSLG_DEMOD_VARIANT=2 CWSLG_PERSIST_WGS_PER_CU=4 run v2-4
CWSLG_DEMOD_VARIANT=2 CWSLG_PERSIST_WGS_PER_CU=8 run v2-8
CWSLG_DEMOD_VARIANT=1 run v1
done
