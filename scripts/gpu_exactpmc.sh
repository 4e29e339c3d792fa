#!/bin/bash
# PMC counters of the exact-mode demod kernel (512 slots, no sync stage)
O=$GRAFT_REPO_ROOT/gpurun_out/exactpmc; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
run() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- python3 $R/bench.py --slots 512 --exact --sync 0 --steps 2 --warmup 1 --no-cpu-baseline --verify 0 > $O/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE GRBM_COUNT
cd $R
python3 - <<'PY'
import csv,glob,collections,os
O=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','exactpmc')
for d in sorted(glob.glob(O+'/*/')):
    for f in glob.glob(d+'**/*counter_collection.csv',recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name'][:40]][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in agg.items():
            if 'exact' in k:
                print(k, {c:'%.4g'%(sum(x)/len(x)) for c,x in v.items()})
PY
rm -rf $O/sq1 $O/sq2
