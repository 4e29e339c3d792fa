#!/bin/bash
# Descriptor upload by copy kernel (default) against hipMemcpyAsync (CWSLG_UPLOAD=dma): parity subset, default bench A/B, timeline gaps
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_demod.py tests/test_gpu_sync.py tests/test_gpu_lifecycle.py tests/test_gpu_longsync.py tests/test_gpu_tune.py -x -q -m gpu 2>&1 | tail -2
for u in dma kernel dma kernel; do
  CWSLG_UPLOAD=$u timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --verify 0 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('4096 slots upload=$u: step %.3f ms (demod %.3f fin %.3f sync %.3f => kernels %.3f) whole %.4f'%(d['ms_per_step'], r['avg_launch_ms'], r['finalize_avg_ms'], r['sync_avg_ms'], r['avg_launch_ms']+r['finalize_avg_ms']+r['sync_avg_ms'], r['whole_path_frac']))"
done
for u in dma kernel; do
  CWSLG_UPLOAD=$u timeout 300 python bench.py --slots 512 --steps 30 --warmup 3 --no-cpu-baseline --verify 0 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('512 slots upload=$u: step %.3f ms (kernels %.3f) whole %.4f'%(d['ms_per_step'], r['avg_launch_ms']+r['finalize_avg_ms']+r['sync_avg_ms'], r['whole_path_frac']))"
done
bash scripts/gpu_gaps.sh 2>&1 | tail -11
