#!/bin/bash
# config 5 (128 WSPR + 128 FST4W-120): long-sync parity tests, kernel stats, run line.  Every step under its own timeout.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; export TMPDIR=/tmp
cd $R
timeout 300 python -m pytest tests/test_gpu_longsync.py -x -q 2>&1 | tail -4
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python3 $R/scripts/run_configs.py --config 5 --steps 2 --verify 1 > $O/c5.json 2>/dev/null
f=$(find $O/c5 -name "*kernel_stats.csv" 2>/dev/null | head -1)
[ -n "$f" ] && grep -v "synth\|phasor\|rocclr" $f | head -${LINES_C5:-12} | cut -c1-150 && cp $f $O/c5_kernel_stats.csv
rm -rf $O/c5
tail -c 1200 $O/c5.json | cut -c1-800
