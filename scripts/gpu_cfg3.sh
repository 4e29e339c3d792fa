#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
python scripts/run_configs.py --config 3 --steps 3 > gpurun_out/config3.json 2> gpurun_out/config3.err; echo "config3 rc=$?"; tail -c 1800 gpurun_out/config3.json; tail -3 gpurun_out/config3.err
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3 -- python3 $R/scripts/run_configs.py --config 3 --steps 2 --verify 0 > $R/gpurun_out/prof3.log 2>&1
cd $R; f=$(find gpurun_out/prof3 -name '*kernel_stats.csv' | head -1); cp $f gpurun_out/kernel_stats_config3.csv; cut -c1-170 gpurun_out/kernel_stats_config3.csv | grep -v -E "rocclr|at::native|synth"
