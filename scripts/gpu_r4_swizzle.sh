#!/bin/bash
# Round 4, item 5: the fast kernel's LDS bank conflicts.  Same-box A/B of demod_kernel<16> with the conflict-free mix-lane order (product) against
# round 3's order (-DCWSLG_MIX_SWIZZLE=0, built here), timing + the two LDS counters.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_demod.py tests/test_gpu_golden.py tests/test_gpu_adversarial.py -x -q -m gpu -k "fast or not exact" 2>&1 | tail -3
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -shared -fPIC -Wno-unused-value -Wno-unused-result -fno-slp-vectorize -DCWSLG_MIX_SWIZZLE=0 \
   -o $O/libcwslgpu_noswz.so cwsl_digi_amd/csrc/cwsl_gpu.hip -ldl 2> $O/noswz_build.log || tail -5 $O/noswz_build.log
for rep in 1 2; do
for cfg in "swz|$R/cwsl_digi_amd/lib/libcwslgpu.so" "noswz|$O/libcwslgpu_noswz.so"; do
  IFS='|' read label lib <<< "$cfg"
  for slots in 512 4096; do
    f=$O/r4_swz_${label}_${slots}.json
    CWSLG_LIB=$lib timeout 300 python3 bench.py --slots $slots --fast --primary-only --sync 0 --steps 10 --warmup 3 --no-cpu-baseline --verify 4 > $f 2> $f.err || tail -5 $f.err
    python3 - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d["roofline"]
print("%-6s %5s slots: %s  demod avg_launch %.3f ms frac %.4f verify %s" % ("$label", "$slots", r["kernel"], r["avg_launch_ms"], r["frac"], d.get("verify", {}).get("max_rel_err")))
PY
  done
done
done
cd /tmp
for cfg in "swz|$R/cwsl_digi_amd/lib/libcwslgpu.so" "noswz|$O/libcwslgpu_noswz.so"; do
  IFS='|' read label lib <<< "$cfg"
  export CWSLG_LIB=$lib
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_swz_$label -- python3 $R/bench.py --slots 512 --fast --primary-only --sync 0 --steps 2 --warmup 1 --no-cpu-baseline --verify 0 > $O/pmc_swz_$label.log 2>&1
  python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_swz_$label/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "demod_kernel" in row["Kernel_Name"]:
            agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in agg.items():
    c={n: sum(x)/len(x) for n,x in v.items()}
    print("$label", k[:50], {n: "%.4g" % x for n,x in c.items()}, "conflict/active = %.3f" % (c.get("SQ_LDS_BANK_CONFLICT",0)/max(1,c.get("SQ_LDS_IDX_ACTIVE",1))))
PY
  rm -rf $O/pmc_swz_$label
done
