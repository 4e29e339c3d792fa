"""Diagnostic (-DCWSLG_STAMP -DCWSLG_STAMP_WAVES build): how far apart the four waves of a demod_exact3_kernel workgroup finish their FIR
(s_memtime at the end of the FIR of the workgroup's 100th tile, one stamp per wave; slot 3 = wave 0 at the start of that FIR)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CWSLG_LIB"] = os.path.join(ROOT, "cwsl_digi_amd/lib/libcwslgpu_stampw.so")
import cwsl_digi_amd as P
ctx = P.Context(0)
S, N, BLK = 512, 2880000, 2048
rb = N // BLK + 3
for s in range(S):
    rx = ctx.receiver_open(192000, BLK, 0, ring_blocks=rb)
    cap = rb * BLK
    ctx.push_synth(rx, s, cap // 2, BLK); ctx.push_synth(rx, s, cap - cap // 2, BLK)
    ctx.channel_open(rx, -90000 + (s * 4373) % 176000, "FT8")
ctx.slot_boundary("FT8", 1)
for k in range(3):
    ctx.ring_commit_all(N, BLK); ctx.process(); ctx.slot_boundary("FT8", 2 + k)
ctx.synchronize()
n = 65536
buf = np.zeros(8 * n, np.uint64)
assert ctx.L.cwslg_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(8 * n)) == 0
st = buf.reshape(n, 8).astype(np.int64)
st = st[(st[:, 4:8] > 0).all(axis=1)]
start = st[:, 3]
ends = st[:, 4:8] - start[:, None]
print("workgroups", len(st))
print("FIR duration by wave (median ticks):", np.median(ends, axis=0).astype(int))
spread = ends.max(axis=1) - ends.min(axis=1)
print("last - first wave to finish (ticks): median %d  p90 %d  max %d" % (np.median(spread), np.percentile(spread, 90), spread.max()))
order = np.argsort(ends, axis=1)
print("which wave finishes last (share):", np.bincount(order[:, -1], minlength=4) / len(st))
