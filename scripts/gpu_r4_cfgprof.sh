#!/bin/bash
# Kernel breakdown (rocprofv3 --stats) of BASELINE configs[4] / configs[2] at full size; parity of the 120 s modes first.
O=$GRAFT_REPO_ROOT/gpurun_out; R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_longsync.py tests/test_gpu_realtime.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python3 scripts/run_configs.py --config 5 --steps 2 > $O/r4_config5b.json 2> $O/r4_config5b.err; tail -c 700 $O/r4_config5b.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5prof -- python3 $R/scripts/run_configs.py --config 5 --steps 2 --verify 0 > $O/c5prof.log 2>&1
cp $(find $O/c5prof -name '*kernel_stats.csv' | head -1) $O/r4_c5_kernel_stats.csv; rm -rf $O/c5prof
cut -c1-150 $O/r4_c5_kernel_stats.csv | grep -i "fftb\|wspr\|fst4w" | head -20
