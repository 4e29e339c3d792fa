#!/bin/bash
# SQ counters of the sync kernels (bench, 4096 slots, fast mode only).  Usage: gpu_syncpmc.sh [env assignments, e.g. CWSLG_LIB=lab CWSLG_SYNC_VARIANT=128]
mkdir -p gpurun_out/pmcs; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for a in "$@"; do export "$a"; done
run() { name=$1; shift
  rm -rf $R/gpurun_out/pmcs/$name
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcs/$name -- python3 $R/bench.py --steps 2 --warmup 1 --fast-only --no-cpu-baseline --verify 0 > $R/gpurun_out/pmcs/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM
run sq3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
cd $R
python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmcs/*/')):
    for f in glob.glob(d+'**/*counter_collection.csv',recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name'][:40]][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in agg.items():
            if 'spectra' in k or 'sync' in k or 'candidates' in k:
                print(d.split('/')[-2], k, {c:'%.4g'%(sum(x)/len(x)) for c,x in v.items()})
PY
