#!/bin/bash
# Kernel breakdown (rocprofv3 --stats) of BASELINE configs[4] and configs[2] at full size.
O=$GRAFT_REPO_ROOT/gpurun_out; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; mkdir -p $O
for c in 5 3; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c${c}prof -- python3 $R/scripts/run_configs.py --config $c --steps 2 --verify 0 > $O/c${c}prof.log 2>&1
cp $(find $O/c${c}prof -name '*kernel_stats.csv' | head -1) $O/r4_c${c}_kernel_stats.csv; rm -rf $O/c${c}prof
done
cut -c1-150 $O/r4_c5_kernel_stats.csv | grep -i "fftb\|wspr\|fst4w" | head -20
