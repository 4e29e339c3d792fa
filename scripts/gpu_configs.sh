#!/bin/bash
# full-size BASELINE configs[2] and configs[4] on one GPU; JSON lines land in gpurun_out/
mkdir -p gpurun_out
python scripts/run_configs.py --config 3 --steps 3 > gpurun_out/config3.json 2> gpurun_out/config3.err; echo "config3 rc=$?"; tail -c 1500 gpurun_out/config3.json; tail -3 gpurun_out/config3.err
python scripts/run_configs.py --config 5 --steps 2 > gpurun_out/config5.json 2> gpurun_out/config5.err; echo "config5 rc=$?"; tail -c 1500 gpurun_out/config5.json; tail -3 gpurun_out/config5.err
