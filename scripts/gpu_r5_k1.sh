#!/bin/bash
# Round 5 step A: K = 1 f32 MFMA as the exact mode's multiplier -- bits and rates (scripts/micro/mfma_k1.hip).
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT/scripts/micro
python3 gen_mfma_k1_loop.py mfma_k1_loop.inc && hipcc --offload-arch=gfx950 -O3 -Wno-unused-value mfma_k1.hip -o mfma_k1 || exit 1
timeout 600 ./mfma_k1 ${1:-512} > $O/r5_mfma_k1.txt 2>&1
cat $O/r5_mfma_k1.txt
