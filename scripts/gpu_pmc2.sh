#!/bin/bash
mkdir -p gpurun_out/pmc; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
run() { name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc/$name -- python3 $R/bench.py --slots ${SLOTS:-512} --steps 2 --warmup 1 --no-cpu-baseline --verify 0 > $R/gpurun_out/pmc/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE GRBM_COUNT
cd $R
python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmc/*/')):
    for f in glob.glob(d+'**/*counter_collection.csv',recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name'][:34]][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in agg.items():
            if 'cwslg' in k and 'synth' not in k and 'phasor' not in k:
                print(d.split('/')[-2], k, {c:'%.4g'%(sum(x)/len(x)) for c,x in v.items()})
PY
