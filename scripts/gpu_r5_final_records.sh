#!/bin/bash
# Round 5, final library: the secondary records again -- sample rates, BASELINE configs[2] / configs[4] at full size, the reference topology (32 receivers x 128 channels).
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
bash scripts/gpu_r5_rates.sh > /dev/null 2>&1
timeout 900 python3 scripts/run_configs.py --config 3 > $O/r5_config3_1024mixed.json 2> $O/r5_config3.err || tail -3 $O/r5_config3.err
timeout 1200 python3 scripts/run_configs.py --config 5 > $O/r5_config5_256long.json 2> $O/r5_config5.err || tail -3 $O/r5_config5.err
timeout 600 python3 bench.py --channels-per-rx 128 --no-cpu-baseline > $O/r5_bench_shared_32x128.json 2> $O/r5_shared.err || tail -3 $O/r5_shared.err
cat $O/r5_rates.txt; tail -c 600 $O/r5_config3_1024mixed.json; echo; tail -c 600 $O/r5_config5_256long.json; echo; tail -c 400 $O/r5_bench_shared_32x128.json
