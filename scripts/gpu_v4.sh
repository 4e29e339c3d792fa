#!/bin/bash
CWSLG_DEMOD_VARIANT=6 python -m pytest tests/test_gpu_demod.py tests/test_gpu_golden.py tests/test_gpu_properties.py tests/test_gpu_lifecycle.py -x -q 2>&1 | tail -3
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --sync 0 --verify 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 demod %.3f ms frac %.3f err %.2e mism %d' % (r['avg_launch_ms'], r['frac'], d['verify']['max_rel_err'], d['verify']['int16_mismatches']))"; }
for rep in 1 2; do
CWSLG_DEMOD_VARIANT=0 run v0
CWSLG_DEMOD_VARIANT=3 run v3
CWSLG_DEMOD_VARIANT=4 run v4
CWSLG_DEMOD_VARIANT=6 run v6
CWSLG_DEMOD_VARIANT=7 run v7
done
