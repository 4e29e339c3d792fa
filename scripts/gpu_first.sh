#!/bin/bash
# First GPU contact: parity tests, smoke, a short bench, a kernel-trace profile.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -8 gpurun_out/smoke.log
timeout 600 python bench.py --slots 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench64.log 2>&1; tail -3 gpurun_out/bench64.log
timeout 900 python bench.py --slots 512 --steps 3 --warmup 1 --cpu-seconds 10 > gpurun_out/bench512.log 2>&1; tail -3 gpurun_out/bench512.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -- python3 $GRAFT_REPO_ROOT/bench.py --slots 512 --steps 3 --warmup 1 --no-cpu-baseline --verify 0 > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/prof1 -name '*stats*' | head; 
