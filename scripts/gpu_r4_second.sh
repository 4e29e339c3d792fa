#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 120 scripts/micro/pk_issue > $O/r4_pk_issue.txt 2>&1; cat $O/r4_pk_issue.txt
timeout 1500 python -m pytest tests/test_gpu_realtime.py tests/test_gpu_demod.py tests/test_gpu_dist.py tests/test_gpu_exact.py tests/test_gpu_skimmer.py tests/test_gpu_adversarial.py tests/test_gpu_longsync.py tests/test_gpu_sync.py tests/test_gpu_ft4sync.py -x -q -m gpu 2>&1 | tail -15 > $O/r4_second_tests.log
tail -5 $O/r4_second_tests.log
