#!/bin/bash
python -m pytest tests/test_gpu_sync.py tests/test_gpu_configs.py -x -q 2>&1 | tail -4
python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value %.0f ms/step %.3f demod %.3f frac %.3f sync %.3f fin %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['sync_avg_ms'], r['finalize_avg_ms']))"
