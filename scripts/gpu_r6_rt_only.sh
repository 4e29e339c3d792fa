#!/bin/bash
cd $GRAFT_REPO_ROOT
# only the realtime part of gpu_r6_records.sh (up to the configs), then the new lifecycle test
sed -n '1,/^PY$/p' scripts/gpu_r6_records.sh > /tmp/rt_part.sh
bash /tmp/rt_part.sh
timeout 600 python -m pytest tests/test_gpu_lifecycle.py -q -m gpu -k "threshold" 2>&1 | tail -3
