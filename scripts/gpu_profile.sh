#!/bin/bash
# tests + bench (with CPU baseline) + rocprofv3 kernel-trace stats of the same bench command
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 3000 gpurun_out/bench_default.json
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --no-cpu-baseline --verify 0 > $R/gpurun_out/prof.log 2>&1
cd $R; f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1); cp $f gpurun_out/kernel_stats.csv; cat gpurun_out/kernel_stats.csv | cut -c1-200
tail -1 gpurun_out/prof.log | cut -c1-400
