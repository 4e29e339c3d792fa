run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --sync 0 --verify 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 %.3f ms' % (r['avg_launch_ms']))"; }
CWSLG_DEMOD_VARIANT=11 run "probe+lds        "
CWSLG_DEMOD_VARIANT=12 run "probe+lds+2k idle"
CWSLG_DEMOD_VARIANT=13 run "probe+lds+4k idle"
CWSLG_DEMOD_VARIANT=14 run "probe+lds+8k idle"
CWSLG_DEMOD_VARIANT=0 run "demod_kernel     "
