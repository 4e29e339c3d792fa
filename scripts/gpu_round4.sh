#!/bin/bash
# Round-4 GPU pass: (1) the -m gpu suite, (2) end-to-end candidate report, (3) the default bench line (4096 FT8 slots on one GPU: the exact-mode
# headline record, the fast-mode record, the CPU baseline), 512- and 64-slot points, the reference topology (32 receivers x 128 channels),
# (4) rocprofv3 kernel-trace stats of the default command, (5) PMC passes (separate runs; HBM traffic + SQ counters), (6) the wall-clock-paced
# ingest harness at north-star scale (-> r4_realtime.json), (7) BASELINE configs[2] / configs[4] at full size, (8) the issue-rate micro-benchmark.
# Everything lands in gpurun_out/r4/; summaries are copied to profiles/ by hand (scripts/make_traffic_json.py regenerates the traffic file).
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd $R
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout 2400 python -m pytest tests -m gpu -q --durations=10 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
  grep -n "passed\|failed\|rc=" $O/pytest.log | tail -4
  timeout 900 python scripts/e2e_report.py 16 > $O/e2e.log 2>&1; cp gpurun_out/e2e_candidates.json $O/ 2>/dev/null
fi
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --slots 512 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench512.json 2> $O/bench512.err
timeout 600 python bench.py --slots 64 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench64.json 2> $O/bench64.err
timeout 600 python bench.py --channels-per-rx 128 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_shared_32x128.json 2> $O/bench_shared_32x128.err
timeout 300 python bench.py --gpus 2 --same-device --dist-backend gloo --slots 512 --steps 5 --warmup 2 --no-cpu-baseline --verify 2 > $O/bench_2ranks_1gpu.json 2> $O/bench_2ranks_1gpu.err
timeout 120 scripts/micro/pk_issue > $O/pk_issue.txt 2>&1
timeout 300 python3 scripts/gpu_rates_exact.py > $O/rates.json 2> $O/rates.err      # both modes at 48 / 96 / 192 kHz, 512 slots, demod launch only
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
              "late worst ms", v["push_late_ms_worst"], "boundaries", [(b["frames_ready_ms"], b["all_frames_fetched_ms"]) for b in v["boundaries"]])
PY
timeout 900 python3 scripts/run_configs.py --config 3 --steps 3 > $O/config3.json 2> $O/config3.err
timeout 900 python3 scripts/run_configs.py --config 5 --steps 2 > $O/config5.json 2> $O/config5.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --verify 0 > $O/stats.log 2>&1
run() { name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --verify 0 > $O/pmc_$name.log 2>&1
}
if [ "${SKIP_PMC:-0}" != 1 ]; then
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE GRBM_COUNT
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM
fi
cd $R
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
python3 - <<'PY'
import csv,glob,collections,json,os
O=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','r4')
tot=collections.defaultdict(dict)
for d in sorted(glob.glob(O+'/pmc_*/')):
    for f in glob.glob(d+'**/*counter_collection.csv',recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name'].split('(')[0]][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in agg.items():
            if 'cwslg' in k and 'synth' not in k and 'phasor' not in k:
                for c,x in v.items():
                    tot[k][c]=sum(x)/len(x)
with open(O+'/pmc_summary.txt','w') as fh:
    for k in sorted(tot):
        fh.write(k+'\n')
        for c in sorted(tot[k]): fh.write('    %-26s %.6g\n'%(c,tot[k][c]))
print(open(O+'/pmc_summary.txt').read())
PY
# drop the bulky raw traces, keep the summaries
rm -rf $O/stats $O/pmc_*/
cut -c1-170 $O/kernel_stats.csv | head -14
tail -c 6000 $O/bench_default.json
