#!/bin/bash
# Randomised soak against the oracle chain with the round's final libraries: SECONDS / SEED from the environment (default 20 min, seed 23).
O=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
S=${SOAK_SECONDS:-1200}; timeout $((S + 300)) python3 scripts/gpu_soak.py --seconds $S --seed ${SOAK_SEED:-23} > $O/r4_soak_long.json 2> $O/r4_soak_long.err; tail -c 700 $O/r4_soak_long.json; tail -2 $O/r4_soak_long.err
