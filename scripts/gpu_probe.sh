#!/bin/bash
CWSLG_DEMOD_VARIANT=5 python -m pytest tests/test_gpu_demod.py tests/test_gpu_golden.py tests/test_gpu_properties.py -x -q 2>&1 | tail -2
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --sync 0 --verify 2 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 demod %.3f ms  -> %.0f GB/s frac %.3f err %.1e' % (r['avg_launch_ms'], r['achieved'], r['frac'], d['verify']['max_rel_err']))"; }
for rep in 1 2; do
CWSLG_DEMOD_VARIANT=0 run v0
CWSLG_DEMOD_VARIANT=5 run v5
CWSLG_DEMOD_VARIANT=6 run v6
done
