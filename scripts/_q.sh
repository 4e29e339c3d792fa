timeout 900 python -m pytest tests/test_gpu_sync.py tests/test_gpu_e2e_candidates.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -2
python3 scripts/gpu_stamps_chan.py 2>&1 | tail -4
for i in 1 2; do python3 bench.py --slots 4096 --fast-only --steps 10 --warmup 3 --no-cpu-baseline --verify 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.3f demod %.3f sync %.3f whole %.4f' % (d['ms_per_step'], r['avg_launch_ms'], r['sync_avg_ms'], r['whole_path_frac']))"; done
