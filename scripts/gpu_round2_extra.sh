#!/bin/bash
# Round-2 side measurements: full-size configs[2] / configs[4], exact mode, 512-slot point, reference topology, PCIe-inclusive rate.
O=$GRAFT_REPO_ROOT/gpurun_out/r2x; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 400 python scripts/run_configs.py --config 3 --steps 3 > $O/config3.json 2> $O/config3.err; tail -c 600 $O/config3.json
timeout 400 python scripts/run_configs.py --config 5 --steps 2 > $O/config5.json 2> $O/config5.err; tail -c 900 $O/config5.json
timeout 300 python bench.py --slots 512 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench512.json 2>/dev/null
timeout 300 python bench.py --slots 512 --exact --sync 0 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench512_exact.json 2>/dev/null
timeout 300 python bench.py --slots 64 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench64.json 2>/dev/null
timeout 300 python bench.py --slots 512 --channels-per-rx 8 --sync 0 --steps 10 --warmup 3 --no-cpu-baseline --verify 0 > $O/bench512_shared8.json 2>/dev/null
timeout 300 python scripts/gpu_h2d.py > $O/h2d.log 2>&1; tail -4 $O/h2d.log
python3 - <<'PY'
import json,os
O=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','r2x')
for f in ('bench512','bench512_exact','bench64','bench512_shared8'):
    try:
        d=json.load(open(os.path.join(O,f+'.json'))); r=d['roofline']
        print('%-18s value %.0f Msps step %.3f ms demod %.3f (frac %.4f) fin %.3f sync %.3f whole %.4f verify %s'%(f,d['value'],d['ms_per_step'],r['avg_launch_ms'],r['frac'],r['finalize_avg_ms'],r['sync_avg_ms'],r['whole_path_frac'],d.get('verify')))
    except Exception as e: print(f,'failed',e)
PY
