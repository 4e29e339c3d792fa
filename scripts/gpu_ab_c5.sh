#!/bin/bash
# Same-box A/B of library builds on BASELINE configs[4] (128 WSPR + 128 FST4W-120 slots): long-sync stage per boundary.  usage: gpu_ab_c5.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
for rep in 1 2; do
for lib in "$@"; do
CWSLG_LIB=$GRAFT_REPO_ROOT/$lib timeout 600 python3 scripts/run_configs.py --config 5 --steps 2 > gpurun_out/r4/abc5.json 2>gpurun_out/r4/abc5.err; python3 -c "
import json; c=json.loads(open('gpurun_out/r4/abc5.json').read().strip().splitlines()[-1]); print('$lib', 'step %.3f demod %.3f long-sync %.3f' % (c['ms_per_step'], c['demod_ms_per_launch'], c['sync_ms_per_boundary']), c['verify']['candidate_lists_identical'], c['verify']['long_mode_e2e_same_frequencies'])" || tail -3 gpurun_out/r4/abc5.err
done; done
