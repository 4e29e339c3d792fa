#!/bin/bash
# Round 4, item 5: which phase of demod_kernel<16> owns its LDS bank-conflict cycles?  Diagnostic builds without one LDS phase each
# (-DCWSLG_DIAG_LDS=1 mix writes / 2 FIR reads / 3 output staging; results are garbage, counters and times are the point), 512 slots.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
for v in 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -shared -fPIC -Wno-unused-value -Wno-unused-result -fno-slp-vectorize -DCWSLG_DIAG_LDS=$v \
     -o $O/libcwslgpu_diag$v.so cwsl_digi_amd/csrc/cwsl_gpu.hip -ldl 2> $O/diag_build$v.log &
done
wait
cd /tmp
for cfg in "full|$R/cwsl_digi_amd/lib/libcwslgpu.so" "no-mix-writes|$O/libcwslgpu_diag1.so" "no-fir-reads|$O/libcwslgpu_diag2.so" "no-staging|$O/libcwslgpu_diag3.so"; do
  IFS='|' read label lib <<< "$cfg"
  export CWSLG_LIB=$lib
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_diag -- python3 $R/bench.py --slots 512 --fast --primary-only --sync 0 --steps 2 --warmup 1 --no-cpu-baseline --verify 0 > $O/pmc_diag.log 2>&1
  python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_diag/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "demod_kernel" in row["Kernel_Name"]:
            agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in agg.items():
    c={n: sum(x)/len(x) for n,x in v.items()}
    print("%-14s" % "$label", {n: "%.4g" % x for n,x in sorted(c.items())}, "conflict/active = %.3f" % (c.get("SQ_LDS_BANK_CONFLICT",0)/max(1,c.get("SQ_LDS_IDX_ACTIVE",1))))
PY
  rm -rf $O/pmc_diag
  timeout 300 python3 $R/bench.py --slots 512 --fast --primary-only --sync 0 --steps 10 --warmup 3 --no-cpu-baseline --verify 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-14s demod avg_launch %.3f ms' % ('$label', d['roofline']['avg_launch_ms']))"
done
rm -f $O/libcwslgpu_diag*.so $O/libcwslgpu_noswz.so
