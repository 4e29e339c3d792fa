#!/bin/bash
# Round 6, first contact: the product library as round 5 left it.  (1) the paced ingest harness at north-star scale on the round-5 exact kernel
# (-> r6_realtime_before.json), (2) demod launch time per sample rate at 4096 slots (-> r6_rates.json), (3) the default bench line.
O=$GRAFT_REPO_ROOT/gpurun_out/r6a; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
RT=cwsl_digi_amd/bin/cwsl_gpu_realtime
timeout 200 $RT --receivers 4096 --channels-per-rx 1 --speed 1 --slots 2 --mode batch > $O/rt_4096x1_batch_x1.json 2> $O/rt_4096x1_batch_x1.err
timeout 200 $RT --receivers 32 --channels-per-rx 128 --speed 1 --slots 2 --mode threads > $O/rt_32x128_x1.json 2> $O/rt_32x128_x1.err
timeout 200 $RT --receivers 32 --channels-per-rx 128 --speed 8 --slots 3 --mode threads > $O/rt_32x128_x8.json 2> $O/rt_32x128_x8.err
timeout 200 $RT --receivers 32 --channels-per-rx 128 --speed 1 --slots 2 --mode threads --process-ms 100 > $O/rt_32x128_x1_p100.json 2> $O/rt_32x128_x1_p100.err
timeout 200 $RT --receivers 4096 --channels-per-rx 1 --speed 1 --slots 2 --mode batch --process-ms 100 > $O/rt_4096x1_batch_x1_p100.json 2> $O/rt_4096x1_batch_x1_p100.err
python3 - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r6a")
out = {}
for name in ("rt_4096x1_batch_x1", "rt_32x128_x1", "rt_32x128_x8", "rt_32x128_x1_p100", "rt_4096x1_batch_x1_p100"):
    try:
        out[name] = json.loads(open(os.path.join(O, name + ".json")).read().strip().splitlines()[-1])
    except Exception as e:
        out[name] = {"error": str(e), "stderr": open(os.path.join(O, name + ".err")).read()[-500:]}
json.dump(out, open(os.path.join(O, "realtime.json"), "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict) and "boundaries" in v:
        print(k, "dropped", v["blocks_dropped"], "gpu busy", v["gpu_busy_fraction"], "launches", v["demod_launches"], "demod_ms", v["demod_ms"],
              "boundaries", [(b["boundary_call_ms"], b["frames_ready_ms"], b["all_frames_fetched_ms"]) for b in v["boundaries"]])
    else: print(k, v)
PY
timeout 900 python3 scripts/gpu_rates_exact.py --slots 4096 > $O/rates4096.json 2> $O/rates4096.err; cat $O/rates4096.err | tail -8
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
