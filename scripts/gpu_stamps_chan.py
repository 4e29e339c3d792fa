"""Diagnostic: one band of ft8_sync_chan_kernel (band 1 of every channel, wave 0; s_memtime stamps of a -DCWSLG_STAMP -DCWSLG_STAMP_SYNC lab build)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CWSLG_LIB"] = os.path.join(ROOT, "cwsl_digi_amd/lib/libcwslgpu_stamp.so")
os.environ["CWSLG_SYNC_VARIANT"] = "0"
import cwsl_digi_amd as P
ctx = P.Context(0)
ctx.set_exact(False)
ctx.enable_sync(True, 1.5, 200, 200, 3000)
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
st = buf.reshape(n, 8).astype(np.int64)[:S]
seq = st[:, [0, 1, 6, 7]]
d = np.diff(seq, axis=1)
ok = (d > 0).all(axis=1) & (d < 10 ** 7).all(axis=1)
d = d[ok]
for k, nme in enumerate(["prefetch issue (24 loads per lane)", "search: 4 bins of wave 0", "wait for the other waves + slide + transposing writes + 3 barriers"]):
    print("  %-70s %8.0f %8.0f" % (nme, np.median(d[:, k]), d[:, k].mean()))
print("  band total", np.median(seq[ok, 3] - seq[ok, 0]))
