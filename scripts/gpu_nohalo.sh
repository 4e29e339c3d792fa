#!/bin/bash
# Upper bound of what a streaming form of demod_kernel (one workgroup walking a channel's tiles with the 31-block halo kept in LDS)
# could gain: the tile kernel and the traffic-only probe WITHOUT the halo's loads and mix (-DCWSLG_DIAG_NOHALO=1 build of the lab
# library: timing only, results are garbage), against the complete ones, same box.  Build both libraries first (see README).
for cfg in "demod_kernel|lab|0" "demod_kernel no halo|cwsl_digi_amd/lib/libcwslgpu_nohalo.so|0" "probe|lab|9" "probe no halo|cwsl_digi_amd/lib/libcwslgpu_nohalo.so|9" "demod_kernel|lab|0" "demod_kernel no halo|cwsl_digi_amd/lib/libcwslgpu_nohalo.so|0"; do
  IFS='|' read label lib v <<< "$cfg"
  CWSLG_LIB=$lib CWSLG_DEMOD_VARIANT=$v timeout 300 python3 bench.py --slots ${S:-512} --sync 0 --fast-only --steps 20 --warmup 3 --no-cpu-baseline --verify 0 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-24s %s  demod %.3f ms' % ('$label', r['kernel'], r['avg_launch_ms']))"
done
