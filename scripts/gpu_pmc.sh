#!/bin/bash
# PMC passes (separate runs; never combined with trace domains other than kernel-trace)
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
run() { # name counters...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc/$name -- python3 $R/bench.py --slots ${SLOTS:-512} --steps 2 --warmup 1 --no-cpu-baseline --verify 0 > $R/gpurun_out/pmc/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM
run sq3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE GRBM_COUNT
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
cd $R
python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmc/*/')):
    for f in glob.glob(d+'**/*counter_collection.csv',recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name'][:40]][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in agg.items():
            if 'demod' in k or 'finalize' in k:
                print(d.split('/')[-2], k, {c:(sum(x)/len(x)) for c,x in v.items()}, 'n=',len(list(v.values())[0]))
PY
