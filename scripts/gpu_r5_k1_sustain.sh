#!/bin/bash
# Round 5 step A, second half: the same instruction mixes looping for seconds, package power and clock sampled beside them -- does moving the
# products to the MFMA instruction change what the chip sustains at its power limit?
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT/scripts/micro
python3 gen_mfma_k1_loop.py mfma_k1_loop.inc && hipcc --offload-arch=gfx950 -O3 -Wno-unused-value mfma_k1.hip -o mfma_k1 || exit 1
{
for v in VALU_ALL BOTH_PK VALU_ALL BOTH_PK BOTH_SC MFMA_ONLY VALU_ONLY; do
  echo "== $v"
  timeout 60 ./mfma_k1 sustain $v 14 &
  BP=$!
  sleep 6
  for k in 1 2 3; do rocm-smi --showpower --showclocks 2>&1 | grep -i "power (W)\|sclk" | sed 's/^.*: //' | tr '\n' ';'; echo; sleep 2; done
  wait $BP
done
} > $O/r5_mfma_k1_sustain.txt 2>&1
cat $O/r5_mfma_k1_sustain.txt
