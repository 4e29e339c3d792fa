#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 1500 python3 scripts/gpu_soak.py --seconds 1200 --seed 23 > $O/r4_soak20.json 2> $O/r4_soak20.err; tail -c 700 $O/r4_soak20.json; tail -2 $O/r4_soak20.err
