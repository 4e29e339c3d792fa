#!/bin/bash
# Round 5: where a step's time goes BETWEEN its kernels (rocprofv3 kernel trace of the default bench, exact record): idle gaps on the device timeline.
O=$GRAFT_REPO_ROOT/gpurun_out/gaps; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --verify 0 --primary-only --steps 10 --warmup 3 > $O/trace.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, os
O = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', 'gaps')
f = glob.glob(O + '/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void cwslg::', '')) for r in csv.DictReader(open(f))]
rows.sort()
# the timed steps: the last 10 launches of the exact demod kernel and everything between them
idx = [i for i, r in enumerate(rows) if 'demod_exact5' in r[2]]
first = idx[-10]
seq = rows[first:]
out = []
tot_gap = {}
for a, b in zip(seq, seq[1:]):
    gap = (b[0] - a[1]) / 1e3
    key = a[2][:28] + ' -> ' + b[2][:28]
    tot_gap.setdefault(key, []).append(gap)
with open(O + '/../r5_gaps.txt', 'w') as fh:
    fh.write('idle time on the device between consecutive kernels of the timed exact steps (us; rocprofv3 kernel trace, default bench, 10 steps)\n')
    for k, v in tot_gap.items():
        fh.write('%-62s n %3d  mean %8.1f  max %8.1f\n' % (k, len(v), sum(v) / len(v), max(v)))
    span = (seq[-1][1] - seq[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in seq) / 1e6
    fh.write('span %.3f ms, kernels %.3f ms, idle %.3f ms (%.2f %%)\n' % (span, busy, span - busy, 100 * (span - busy) / span))
print(open(O + '/../r5_gaps.txt').read())
PY
rm -rf $O
