#!/bin/bash
# Board power and shader clock while the default bench loops (evidence for the power-limited clock of the demod kernel).
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^=\|^$" | head -20 > $O/power_idle.txt
(timeout 120 python bench.py --steps 1500 --warmup 2 --no-cpu-baseline --verify 0 ${BENCH_ARGS:-} > $O/power_bench.json 2>/dev/null) &
BP=$!
sleep 10
for k in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk" | tr '\n' ';' ; echo
  sleep 2
done > $O/power_load.txt
wait $BP
echo "--- idle"; cat $O/power_idle.txt; echo "--- under load"; cat $O/power_load.txt; tail -c 400 $O/power_bench.json
