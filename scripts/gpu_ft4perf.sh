#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ft4sync.py -x -q 2>&1 | tail -2
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3 -- python3 $R/scripts/run_configs.py --config 3 --steps 2 --verify 0 > $R/gpurun_out/prof3.log 2>&1
cd $R; f=$(find gpurun_out/prof3 -name '*kernel_stats.csv' | head -1); cut -c1-150 $f | grep -E "ft4_"
