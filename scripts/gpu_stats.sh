#!/bin/bash
# rocprofv3 kernel-trace stats of the bench command (short)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --no-cpu-baseline --verify 0 --steps 3 --warmup 1 ${BENCH_ARGS} > $R/gpurun_out/prof.log 2>&1
cd $R; f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1); cp $f gpurun_out/kernel_stats.csv; cut -c1-160 gpurun_out/kernel_stats.csv | grep -v -E "rocclr|at::native"
