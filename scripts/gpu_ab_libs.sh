#!/bin/bash
# A/B of alternative builds of the library (cwsl_digi_amd/lib/ab/*.so, selected through CWSLG_LIB) against the default one.
cd $GRAFT_REPO_ROOT
run() { CWSLG_LIB=$1 timeout 300 python bench.py --slots ${S:-512} --steps 10 --warmup 3 --no-cpu-baseline --verify 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-28s step %.3f demod %.3f sync %.3f whole %.4f'%('$2', d['ms_per_step'], r['avg_launch_ms'], r['sync_avg_ms'], r['whole_path_frac']))"; }
for rep in 1 2; do
run "" default
for f in cwsl_digi_amd/lib/ab/*.so; do run $GRAFT_REPO_ROOT/$f $(basename $f); done
done
