#!/bin/bash
# Same-box A/B of CWSLG_SYNC_VARIANT values of the lab library on the default bench's sync stage: usage gpu_ab_variant.sh 0 16 ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
for rep in 1 2; do
for v in "$@"; do
CWSLG_LIB=$GRAFT_REPO_ROOT/cwsl_digi_amd/lib/libcwslgpu_lab.so CWSLG_SYNC_VARIANT=$v timeout 600 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --primary-only --fast --verify 2 > gpurun_out/r4/ab.json 2>gpurun_out/r4/ab.err; python3 -c "
import json,sys; d=json.loads(open('gpurun_out/r4/ab.json').read().strip().splitlines()[-1]); r=d['roofline']
print('variant $v', 'step %.3f demod %.3f sync %.3f spectra %.3f search %.3f' % (d['ms_per_step'], r['avg_launch_ms'], r['sync_avg_ms'], d['roofline_sync']['per_kernel']['spectra']['avg_ms'], d['roofline_sync']['per_kernel']['search']['avg_ms']))" || tail -3 gpurun_out/r4/ab.err
done; done
