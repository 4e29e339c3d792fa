#!/bin/bash
# Round 6: randomised soak against the oracle chain (exact mode: demod_exact5_kernel; FT8 + sync channels finalised inside the spectra kernel): SOAK_SECONDS / SOAK_SEED from the environment.
O=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
S=${SOAK_SECONDS:-900}; timeout $((S + 300)) python3 scripts/gpu_soak.py --seconds $S --seed ${SOAK_SEED:-61} > $O/r6_soak.json 2> $O/r6_soak.err; tail -c 700 $O/r6_soak.json; tail -2 $O/r6_soak.err
