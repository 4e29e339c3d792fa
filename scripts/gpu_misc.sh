#!/bin/bash
# (1) N>1 code path on the 1-GPU box: 2 ranks on GPU 0, gloo for the slot-boundary all-reduce
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --slots 128 --dist-backend gloo --same-device --no-cpu-baseline 2>&1 | tail -2 | cut -c1-700
# (2) the north-star slot count on ONE GPU: 4096 FT8 slots x 15 s = 94 GB of IQ resident
timeout 900 python bench.py --slots 4096 --steps 3 --warmup 1 --no-cpu-baseline --verify 2 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); r=j['roofline']
print('slots',j['config']['slots_per_gpu'],'Msps %.0f'%j['value'],'ms/step %.3f'%j['ms_per_step'],'demod ms %.3f'%r['avg_launch_ms'],'frac %.3f'%r['frac'],'sync ms %.3f'%r['sync_avg_ms'],'rt slots %.0f'%j['realtime_ft8_slots'],'setup %.1f'%j['setup_s'],'verify',j['verify'])
"
timeout 600 python bench.py --slots 64 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); r=j['roofline']
print('slots',j['config']['slots_per_gpu'],'Msps %.0f'%j['value'],'ms/step %.3f'%j['ms_per_step'],'demod ms %.3f'%r['avg_launch_ms'],'frac %.3f'%r['frac'],'sync ms %.3f'%r['sync_avg_ms'])
"
