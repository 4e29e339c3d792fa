#!/bin/bash
# Round-3 GPU pass: (1) the -m gpu suite, (2) end-to-end candidate report, (3) the default bench line (4096 FT8 slots on one GPU: the
# fast-mode record, the exact-mode record and the CPU baseline), (4) rocprofv3 kernel-trace stats of the same command, (5) PMC passes
# (separate runs; HBM traffic + SQ counters), (6) the 512- and 64-slot points.  Everything lands in gpurun_out/r3/; summaries are
# copied to profiles/ by hand.
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd $R
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout 2400 python -m pytest tests -m gpu -q --durations=10 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
  grep -n "passed\|failed\|rc=" $O/pytest.log | tail -4
  timeout 900 python scripts/e2e_report.py 16 > $O/e2e.log 2>&1; cp gpurun_out/e2e_candidates.json $O/ 2>/dev/null
fi
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --slots 512 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench512.json 2> $O/bench512.err
timeout 600 python bench.py --slots 64 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench64.json 2> $O/bench64.err
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
O=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','r3')
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
tail -c 4000 $O/bench_default.json
