#!/bin/bash
# Round profile: (1) default bench line, (2) rocprofv3 kernel-trace stats of the same command,
# (3) PMC passes (separate runs) for HBM traffic + SQ counters.  Everything lands in gpurun_out/round/.
O=$GRAFT_REPO_ROOT/gpurun_out/round; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --verify 0 > $O/stats.log 2>&1
run() { name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --verify 0 > $O/pmc_$name.log 2>&1
}
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE GRBM_COUNT
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
cd $R
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
rm -rf $O/stats
python3 - <<'PY'
import csv,glob,collections,json,os
O=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','round')
rows=[]
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
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2 $O/pmc_tcc      # (the raw traces exceed what gpurun copies back; the summary is what is kept)
tail -c 2500 $O/bench_default.json
