#!/bin/bash
# Round 5 step A, third run: the one-product-buffer schedule (both_sc1) against the two-buffer one.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT/scripts/micro
python3 gen_mfma_k1_loop.py mfma_k1_loop.inc && hipcc --offload-arch=gfx950 -O3 -Wno-unused-value mfma_k1.hip -o mfma_k1 || exit 1
{
timeout 300 ./mfma_k1 8 | grep -v "^GAP\|^SGAP"
for v in BOTH_SC1 BOTH_SC; do
  echo "== $v"
  timeout 60 ./mfma_k1 sustain $v 8 | tail -2 &
  BP=$!
  sleep 5
  rocm-smi --showpower --showclocks 2>&1 | grep -i "power (W)\|sclk" | sed 's/^.*: //' | tr '\n' ';'; echo
  wait $BP
done
} > $O/r5_mfma_k1_b.txt 2>&1
cat $O/r5_mfma_k1_b.txt
