#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q --durations=8 > $O/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/r4_pytest_gpu.log
grep -n "passed\|failed\|rc=" $O/r4_pytest_gpu.log | tail -4
