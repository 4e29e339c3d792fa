#!/bin/bash
# Idle gaps between the kernels of a bench step (rocprofv3 kernel trace timestamps): where step time minus kernel time goes
O=$GRAFT_REPO_ROOT/gpurun_out/gaps; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --verify 0 ${BENCH_ARGS:-} > $O/run.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, os
O=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','gaps')
ev=[]
for f in glob.glob(O+'/t/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-60:]))
for f in glob.glob(O+'/t/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY '+r.get('Direction','')))
ev.sort()
# steps: find demod kernels; take the last 5 demod launches and everything between them
idx=[i for i,e in enumerate(ev) if 'demod_kernel' in e[2]]
print('demod launches', len(idx), 'events', len(ev))
lo=idx[-6]; hi=idx[-1]
seg=ev[lo:hi]
busy=sum(e[1]-e[0] for e in seg)
span=ev[hi][0]-ev[lo][0]
print('5 steps: span %.3f ms, busy %.3f ms, idle %.3f ms (%.2f%%)'%(span/1e6, busy/1e6, (span-busy)/1e6, 100*(span-busy)/span))
# per-step listing of the last step
lo2=idx[-2]
prev_end=None
for s,e,n in ev[lo2:hi+1]:
    gap=(s-prev_end)/1e3 if prev_end else 0.0
    print('  gap %8.1f us  dur %10.1f us  %s'%(gap,(e-s)/1e3,n))
    prev_end=max(prev_end or 0,e)
PY
rm -rf $O/t
