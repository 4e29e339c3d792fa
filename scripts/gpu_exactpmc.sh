#!/bin/bash
mkdir -p gpurun_out/pmcx; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
run() { name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcx/$name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --verify 0 --sync 0 --exact > $R/gpurun_out/pmcx/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM
run sq3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES GRBM_GUI_ACTIVE
cd $R
python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmcx/*/')):
    for f in glob.glob(d+'**/*counter_collection.csv',recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name'][:40]][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in agg.items():
            if 'exact' in k:
                print(d.split('/')[-2], k, {c:'%.4g'%(sum(x)/len(x)) for c,x in v.items()})
PY
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --sync 0 --exact 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('exact demod %.3f ms frac %.3f err %.2e mism %d' % (r['avg_launch_ms'], r['frac'], d['verify']['max_rel_err'], d['verify']['int16_mismatches']))"
