#!/bin/bash
# Round 4, after the last library change (sync buffers allocated at channel open; two radix-2 levels per pass in fftb_stage2): the -m gpu suite,
# the paced ingest harness (-> realtime.json) and configs[4] again.
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -n "passed\|failed\|rc=" $O/pytest.log | tail -3
RT=cwsl_digi_amd/bin/cwsl_gpu_realtime
timeout 300 $RT --receivers 32 --channels-per-rx 128 --speed 1 --slots 3 --mode threads > $O/rt_32x128_x1.json 2> $O/rt_32x128_x1.err
timeout 300 $RT --receivers 32 --channels-per-rx 128 --speed 8 --slots 3 --mode threads > $O/rt_32x128_x8.json 2> $O/rt_32x128_x8.err
timeout 300 $RT --receivers 32 --channels-per-rx 128 --speed 1 --slots 3 --mode threads --process-ms 500 > $O/rt_32x128_x1_p500.json 2> $O/rt_32x128_x1_p500.err
timeout 300 $RT --receivers 4096 --channels-per-rx 1 --speed 1 --slots 2 --mode batch > $O/rt_4096x1_batch_x1.json 2> $O/rt_4096x1_batch_x1.err
timeout 300 $RT --receivers 4096 --channels-per-rx 1 --speed 1 --slots 1 --slot-blocks 470 --mode threads > $O/rt_4096x1_threads_x1.json 2> $O/rt_4096x1_threads_x1.err
timeout 300 $RT --receivers 4096 --channels-per-rx 1 --speed 0 --slots 1 --slot-blocks 400 --mode batch > $O/rt_4096x1_batch_unpaced.json 2> $O/rt_4096x1_batch_unpaced.err
python3 - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r4")
out = {"note": "cwsl_gpu_realtime (csrc/host/realtime_main.cpp): wall-clock-paced pushes through the C ABI, exact mode, FT8 sync stage on; "
               "slot = 1406 blocks of 2048 samples (14.997 s); one discarded partial slot first.  boundaries[k]: ms from the cwslg_slot_boundary "
               "call to (its return / every frame and candidate list final on the device / all 4096 int16 frames in host memory)."}
for name in ("rt_32x128_x1", "rt_32x128_x8", "rt_32x128_x1_p500", "rt_4096x1_batch_x1", "rt_4096x1_threads_x1", "rt_4096x1_batch_unpaced"):
    try:
        out[name] = json.loads(open(os.path.join(O, name + ".json")).read().strip().splitlines()[-1])
    except Exception as e:
        out[name] = {"error": str(e), "stderr": open(os.path.join(O, name + ".err")).read()[-500:]}
json.dump(out, open(os.path.join(O, "realtime.json"), "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict) and "boundaries" in v:
        print(k, "dropped", v["blocks_dropped"], "cpu s/s", v["host_cpu_seconds_per_second"], "gpu busy", v["gpu_busy_fraction"], "H2D GB/s", v["h2d_gbytes_per_s"],
              "late worst ms", v["push_late_ms_worst"], "boundaries", [(b["boundary_call_ms"], b["frames_ready_ms"], b["all_frames_fetched_ms"]) for b in v["boundaries"]])
PY
timeout 900 python3 scripts/run_configs.py --config 5 --steps 2 > $O/config5.json 2> $O/config5.err; tail -c 500 $O/config5.json
timeout 900 python3 scripts/run_configs.py --config 3 --steps 3 > $O/config3.json 2> $O/config3.err; tail -c 400 $O/config3.json
