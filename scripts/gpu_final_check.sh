#!/bin/bash
timeout 900 python bench.py --cpu-seconds 8 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); r=j['roofline']
print('N=1: Msps %.0f'%j['value'],'ms/step %.3f'%j['ms_per_step'],'demod ms %.3f'%r['avg_launch_ms'],'frac %.3f'%r['frac'],'traffic',r['traffic'],'cpu',j['cpu_baseline'],'verify',j['verify'])
"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 3 --warmup 1 --slots 128 --dist-backend gloo --same-device --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
