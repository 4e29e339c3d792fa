"""Diagnostic (-DCWSLG_STAMP -DCWSLG_STAMP_TOPS build): loop-top to loop-top time of eight consecutive tiles of every demod_exact3_kernel workgroup."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CWSLG_LIB"] = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "cwsl_digi_amd/lib/libcwslgpu_stampt.so")
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
st = st[(st > 0).all(axis=1)]
d = np.diff(st, axis=1)
print("workgroups", len(st), " tile period (ticks) by position 96->97 ... 102->103: median", np.median(d, axis=0).astype(int), " mean", d.mean(axis=0).astype(int))
print("all periods: p5 %d p50 %d p95 %d max %d" % tuple(np.percentile(d, [5, 50, 95, 100]).astype(int)))
