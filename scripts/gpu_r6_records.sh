#!/bin/bash
# Round 6, final library: the secondary records -- the paced ingest harness at north-star scale, BASELINE configs[2] / configs[4] at full size, the reference
# topology (32 receivers x 128 channels), candidate lists end to end from IQ, sample rates at 4096 slots.
O=$GRAFT_REPO_ROOT/gpurun_out/r6rec; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
RT=cwsl_digi_amd/bin/cwsl_gpu_realtime
timeout 200 $RT --receivers 4096 --channels-per-rx 1 --speed 1 --slots 2 --mode batch > $O/rt_4096x1_batch_x1.json 2> $O/rt_4096x1_batch_x1.err
timeout 200 $RT --receivers 32 --channels-per-rx 128 --speed 1 --slots 2 --mode threads > $O/rt_32x128_x1.json 2> $O/rt_32x128_x1.err
timeout 200 $RT --receivers 32 --channels-per-rx 128 --speed 8 --slots 3 --mode threads > $O/rt_32x128_x8.json 2> $O/rt_32x128_x8.err
timeout 200 $RT --receivers 4096 --channels-per-rx 1 --speed 1 --slots 2 --mode batch --process-ms 10.6 --process-threshold -1 > $O/rt_4096x1_batch_x1_wake.json 2> $O/rt_4096x1_batch_x1_wake.err
timeout 200 $RT --receivers 4096 --channels-per-rx 1 --speed 1 --slots 2 --mode batch --process-ms 100 > $O/rt_4096x1_batch_x1_p100.json 2> $O/rt_4096x1_batch_x1_p100.err
timeout 200 $RT --receivers 4096 --channels-per-rx 1 --speed 1 --slots 2 --mode batch --flush-before 8 > $O/rt_4096x1_batch_x1_flush8.json 2> $O/rt_4096x1_batch_x1_flush8.err
timeout 200 $RT --receivers 32 --channels-per-rx 128 --speed 1 --slots 2 --mode threads --flush-before 8 > $O/rt_32x128_x1_flush8.json 2> $O/rt_32x128_x1_flush8.err
python3 - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r6rec")
out = {"note": "cwsl_gpu_realtime (csrc/host/realtime_main.cpp) on the round-6 library: wall-clock-paced pushes through the C ABI, exact mode (demod_exact5_kernel<16>), "
               "FT8 sync stage on (the slot's finalise inside symbol_spectra_v2_kernel); slot = 1406 blocks of 2048 samples (14.997 s); one discarded partial slot first.  "
               "boundaries[k]: ms from the cwslg_slot_boundary call to (its return / every frame and candidate list final on the device / all 4096 int16 frames in host memory).  "
               "gpu_busy_fraction = sum of the kernels' HIP-event spans / wall.  demod_redundancy = blocks put through the arithmetic (warm-up included) / blocks delivered.  "
               "_wake: cwslg_process() after EVERY block period with the library's launch threshold (cwslg_set_process_threshold(ctx, -1)); _p100: every 100 ms, no threshold "
               "(round 5's behaviour); _flush8: cwslg_flush eight blocks (85 ms) ahead of every boundary -- a host with a slot clock knows when one is due."}
for name in ("rt_4096x1_batch_x1", "rt_32x128_x1", "rt_32x128_x8", "rt_4096x1_batch_x1_wake", "rt_4096x1_batch_x1_p100", "rt_4096x1_batch_x1_flush8", "rt_32x128_x1_flush8"):
    try:
        out[name] = json.loads(open(os.path.join(O, name + ".json")).read().strip().splitlines()[-1])
    except Exception as e:
        out[name] = {"error": str(e), "stderr": open(os.path.join(O, name + ".err")).read()[-500:]}
json.dump(out, open(os.path.join(O, "realtime.json"), "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict) and "boundaries" in v:
        print(k, "dropped", v["blocks_dropped"], "gpu busy", v["gpu_busy_fraction"], "launches", v["demod_launches"], "demod_ms", v["demod_ms"], "redund", v["demod_redundancy"],
              "boundaries", [(b["boundary_call_ms"], b["frames_ready_ms"], b["all_frames_fetched_ms"]) for b in v["boundaries"]])
    elif k != "note": print(k, v)
PY
timeout 900 python3 scripts/run_configs.py --config 3 > $O/config3_1024mixed.json 2> $O/config3.err || tail -3 $O/config3.err
timeout 1200 python3 scripts/run_configs.py --config 5 > $O/config5_256long.json 2> $O/config5.err || tail -3 $O/config5.err
timeout 600 python3 bench.py --channels-per-rx 128 --no-cpu-baseline > $O/bench_shared_32x128.json 2> $O/shared.err || tail -3 $O/shared.err
timeout 900 python3 scripts/e2e_report.py > $O/e2e_candidates.json 2> $O/e2e.err || tail -3 $O/e2e.err
timeout 900 python3 scripts/gpu_rates_exact.py --slots 4096 > $O/rates4096.json 2> $O/rates4096.err; tail -6 $O/rates4096.err
tail -c 600 $O/config3_1024mixed.json; echo; tail -c 600 $O/config5_256long.json; echo; tail -c 400 $O/bench_shared_32x128.json; echo; tail -c 600 $O/e2e_candidates.json
