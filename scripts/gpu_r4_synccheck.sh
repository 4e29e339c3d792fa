#!/bin/bash
# Sync-stage parity tests + the sync timing of the default bench (fast check after a change to the sync kernels).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_sync.py tests/test_gpu_longsync.py tests/test_gpu_ft4sync.py -m gpu -q -x > gpurun_out/r4/pytest_sync.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/pytest_sync.log
grep -n "passed\|failed\|rc=" gpurun_out/r4/pytest_sync.log | tail -3
timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --primary-only > gpurun_out/r4/bench_check.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('gpurun_out/r4/bench_check.json').read().strip().splitlines()[-1]); r=d['roofline']
print('exact', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['sync_avg_ms'], d['roofline_sync']['per_kernel']['spectra']['avg_ms'], d['roofline_sync']['per_kernel']['search']['avg_ms'], d['verify']['max_rel_err'])"
if [ -n "$1" ]; then
timeout 900 python3 scripts/run_configs.py --config 5 --steps 2 > gpurun_out/r4/config5.json 2>/dev/null; python3 -c "
import json; c=json.loads(open('gpurun_out/r4/config5.json').read().strip().splitlines()[-1]); print('config5', c['value'], c['ms_per_step'], c['sync_ms_per_boundary'], c['verify']['candidate_lists_identical'])"
timeout 900 python3 scripts/run_configs.py --config 3 --steps 3 > gpurun_out/r4/config3.json 2>/dev/null; python3 -c "
import json; c=json.loads(open('gpurun_out/r4/config3.json').read().strip().splitlines()[-1]); print('config3', c['value'], c['ms_per_step'], c['sync_ms_per_boundary'], c['verify']['candidate_lists_identical'])"
fi
