#!/bin/bash
# Round 4 extras: (1) the launcher with EIGHT ranks on the one GPU of the box (gloo carries the rendezvous: RCCL refuses several ranks on one device) --
# the shape of the driver's N = 8 run, process group and sharding included; (2) real time at 4x the north-star channel count (128 receivers x 128 channels).
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --gpus 8 --same-device --dist-backend gloo --slots 64 --steps 3 --warmup 1 --no-cpu-baseline --verify 2 > $O/bench_8ranks_1gpu.json 2> $O/bench_8ranks_1gpu.err; echo "rc=$?"; tail -c 1500 $O/bench_8ranks_1gpu.json; tail -3 $O/bench_8ranks_1gpu.err
RT=cwsl_digi_amd/bin/cwsl_gpu_realtime
timeout 300 $RT --receivers 128 --channels-per-rx 128 --speed 1 --slots 2 --mode threads --fetch-threads 8 > $O/rt_128x128_x1.json 2> $O/rt_128x128_x1.err; tail -c 1200 $O/rt_128x128_x1.json; tail -3 $O/rt_128x128_x1.err
