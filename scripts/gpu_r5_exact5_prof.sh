#!/bin/bash
# Round 5: where demod_exact5_kernel's time goes -- per-kernel times (rocprofv3 --stats), SQ counters (separate passes), package power beside a loop.
O=$GRAFT_REPO_ROOT/gpurun_out/x5prof; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
SLOTS=${1:-4096}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --slots $SLOTS --exact --sync 0 --steps 3 --warmup 1 --no-cpu-baseline --verify 0 > $O/stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); echo "== kernel stats"; head -8 $f | cut -c1-200
run() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- python3 $R/bench.py --slots $SLOTS --exact --sync 0 --steps 2 --warmup 1 --no-cpu-baseline --verify 0 > $O/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE GRBM_COUNT
run sq3 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH
cd $R
python3 - <<'PY'
import csv,glob,collections,os
O=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','x5prof')
for d in sorted(glob.glob(O+'/sq*/')):
    for f in glob.glob(d+'**/*counter_collection.csv',recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name'][:40]][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in agg.items():
            if 'exact' in k:
                print(k, {c:'%.4g'%(sum(x)/len(x)) for c,x in v.items()})
PY
rm -rf $O/sq1 $O/sq2 $O/sq3
echo "== power"
timeout 100 python bench.py --slots $SLOTS --exact --sync 0 --steps 600 --warmup 2 --no-cpu-baseline --verify 0 > $O/power.json 2>/dev/null &
BP=$!
sleep 25
for k in 1 2 3 4; do rocm-smi --showpower --showclocks 2>&1 | grep -i "power (W)\|sclk" | sed 's/^.*: //' | tr '\n' ';'; echo; sleep 2; done
wait $BP
