#!/bin/bash
# Round 4, first GPU pass: the new tests, the issue-rate micro-benchmark, the default bench line (exact headline + in-kernel clock),
# and the wall-clock-paced ingest harness at north-star scale.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_realtime.py tests/test_gpu_demod.py tests/test_gpu_dist.py tests/test_gpu_exact.py -x -q -m gpu 2>&1 | tail -15 > $O/r4_first_tests.log
tail -5 $O/r4_first_tests.log
timeout 120 scripts/micro/pk_latency > $O/r4_pk_latency.txt 2>&1; cat $O/r4_pk_latency.txt
timeout 900 python bench.py --steps 5 --warmup 2 > $O/r4_bench4096_first.json 2> $O/r4_bench4096_first.err || tail -5 $O/r4_bench4096_first.err
python3 - <<PY
import json
d=json.loads(open("$O/r4_bench4096_first.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("exact:", d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["whole_path_frac"], r.get("valu_pipe"))
f=d.get("fast"); 
if f: print("fast:", f["value"], f["ms_per_step"], f["roofline"]["avg_launch_ms"], f["roofline"]["frac"], f["roofline"]["whole_path_frac"])
print("cpu:", d.get("cpu_baseline"))
PY
RT=cwsl_digi_amd/bin/cwsl_gpu_realtime
timeout 300 $RT --receivers 32 --channels-per-rx 128 --speed 8 --slots 3 --mode threads > $O/r4_rt_32x128_x8.json 2> $O/r4_rt_32x128_x8.err; tail -c 1500 $O/r4_rt_32x128_x8.json; tail -3 $O/r4_rt_32x128_x8.err
timeout 300 $RT --receivers 32 --channels-per-rx 128 --speed 1 --slots 3 --mode threads > $O/r4_rt_32x128_x1.json 2> $O/r4_rt_32x128_x1.err; tail -c 1500 $O/r4_rt_32x128_x1.json; tail -3 $O/r4_rt_32x128_x1.err
timeout 300 $RT --receivers 4096 --channels-per-rx 1 --speed 1 --slots 2 --mode batch > $O/r4_rt_4096x1_batch_x1.json 2> $O/r4_rt_4096x1_batch_x1.err; tail -c 1500 $O/r4_rt_4096x1_batch_x1.json; tail -3 $O/r4_rt_4096x1_batch_x1.err
timeout 300 $RT --receivers 4096 --channels-per-rx 1 --speed 0 --slots 1 --slot-blocks 400 --mode batch > $O/r4_rt_4096x1_batch_unpaced.json 2> $O/r4_rt_4096x1_batch_unpaced.err; tail -c 1500 $O/r4_rt_4096x1_batch_unpaced.json; tail -3 $O/r4_rt_4096x1_batch_unpaced.err
