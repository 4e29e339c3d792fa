#!/bin/bash
# The demod kernel against the memory system ("Where the demod kernel stands"): the traffic-only probe, the
# default tile kernel and the matrix-core alternative, interleaved on one box.   gpurun -- 'bash scripts/gpu_probe.sh'
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --sync 0 --verify $3 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 %.3f ms per launch -> %.0f GB/s = %.3f of 8 TB/s' % (r['avg_launch_ms'], r['achieved'], r['frac']))"; }
for rep in 1 2; do
CWSLG_DEMOD_VARIANT=9 run "probe (traffic only)      " "" 0
CWSLG_DEMOD_VARIANT=11 run "probe + 40 KB LDS per WG  " "" 0
CWSLG_DEMOD_VARIANT=0 run "demod_kernel (default)    " "" 1
CWSLG_DEMOD_VARIANT=6 run "demod_mfma1p_kernel (6/CU)" "" 1
done
