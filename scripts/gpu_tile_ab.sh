#!/bin/bash
# 192-output tiles at five workgroups per CU (CWSLG_DEMOD_VARIANT=15) against the default 256-output tiles at four
cd $GRAFT_REPO_ROOT
CWSLG_DEMOD_VARIANT=15 timeout 900 python -m pytest tests/test_gpu_demod.py tests/test_gpu_golden.py tests/test_gpu_properties.py tests/test_gpu_lifecycle.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -4
for v in ${VARIANTS:-0 15 0 15}; do
  CWSLG_DEMOD_VARIANT=$v timeout 300 python bench.py --slots 512 --sync 0 --steps 20 --warmup 3 --no-cpu-baseline --verify 8 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('512 slots variant $v: demod %.3f ms frac %.4f verify err %s mism %s'%(r['avg_launch_ms'], r['frac'], d['verify'].get('max_rel_err'), d['verify'].get('int16_mismatches')))"
done
for v in ${VARIANTS:-0 15 0 15}; do
  CWSLG_DEMOD_VARIANT=$v timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --verify 0 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('4096 slots variant $v: step %.3f demod %.3f ms frac %.4f whole %.4f'%(d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['whole_path_frac']))"
done
