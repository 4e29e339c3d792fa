#!/bin/bash
# Randomised soak against the oracle chain with the round's libraries (demod_exact5_kernel in the exact mode): SOAK_SECONDS / SOAK_SEED from the environment.
O=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
S=${SOAK_SECONDS:-900}; timeout $((S + 300)) python3 scripts/gpu_soak.py --seconds $S --seed ${SOAK_SEED:-51} > $O/r5_soak.json 2> $O/r5_soak.err; tail -c 700 $O/r5_soak.json; tail -2 $O/r5_soak.err
