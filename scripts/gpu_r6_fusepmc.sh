#!/bin/bash
# Round 6: what does the fused finalise cost the spectra kernel?  SQ / TCC counters and kernel durations of symbol_spectra_v2_kernel and ft8_sync_chan_kernel
# for two builds (CWSLG_FUSE_FIN_DEFAULT=0 / 1), same box; separate --pmc passes (MI355X_MICROARCH.md).
mkdir -p gpurun_out/pmc6; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
run() { tag=$1; name=$2; shift 2
  rm -rf $R/gpurun_out/pmc6/${tag}_$name
  ( cd /tmp; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc6/${tag}_$name -- python3 $R/bench.py --steps 2 --warmup 1 --fast-only --no-cpu-baseline --verify 0 > $R/gpurun_out/pmc6/${tag}_$name.log 2>&1 )
}
for F in ${FUSES:-0 1}; do
  export CWSLG_HIPCC_EXTRA="-DCWSLG_FUSE_FIN_DEFAULT=$F ${EXTRA:-}"
  ( cd $R; python3 -c "
from cwsl_digi_amd import build as B
B.build(force=True)" > /dev/null 2>&1 ) || { echo "fuse=$F: build failed"; continue; }
  run f$F sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
  run f$F sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
  run f$F tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE
done
unset CWSLG_HIPCC_EXTRA
cd $R
python3 - <<'PY' | tee gpurun_out/pmc6/summary.txt
import csv,glob,collections,os
for d in sorted(glob.glob('gpurun_out/pmc6/*/')):
    tag=d.split('/')[-2]
    for f in glob.glob(d+'**/*counter_collection.csv',recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name'][:34]][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in agg.items():
            if 'spectra' in k or 'sync_chan' in k or 'finalize' in k:
                print(tag, k, {c:'%.4g'%(sum(x)/len(x)) for c,x in sorted(v.items())})
    for f in glob.glob(d+'**/*kernel_trace.csv',recursive=True):
        dur=collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            dur[row['Kernel_Name'][:34]].append((int(row['End_Timestamp'])-int(row['Start_Timestamp']))/1e6)
        for k,v in dur.items():
            if 'spectra' in k or 'sync_chan' in k or 'finalize' in k:
                print(tag, k, 'ms', ['%.3f'%x for x in v])
PY
rm -rf gpurun_out/pmc6/*/      # (the raw traces exceed what gpurun copies back)
python3 -c "
from cwsl_digi_amd import build as B
B.build(force=True)" > /dev/null 2>&1
