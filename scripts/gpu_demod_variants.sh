#!/bin/bash
# demod kernel alternatives (CWSLG_DEMOD_VARIANT), same box: 512 slots without the sync stage, then the default bench
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-0 2 0 2 1 7}; do
  CWSLG_DEMOD_VARIANT=$v timeout 300 python bench.py --slots 512 --sync 0 --steps 20 --warmup 3 --no-cpu-baseline --verify 8 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('512 slots variant $v: demod %.3f ms frac %.4f verify err %s mism %s'%(r['avg_launch_ms'], r['frac'], d['verify'].get('max_rel_err'), d['verify'].get('int16_mismatches')))"
done
for v in ${VARIANTS4096:-0 2 0 2}; do
  CWSLG_DEMOD_VARIANT=$v timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --verify 0 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('4096 slots variant $v: step %.3f demod %.3f ms frac %.4f whole %.4f'%(d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['whole_path_frac']))"
done
