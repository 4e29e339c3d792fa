"""Diagnostic: where a demod tile spends its cycles (s_memtime stamps of the -DCWSLG_STAMP build)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CWSLG_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cwsl_digi_amd", "lib", "libcwslgpu_stamp.so")
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
rc = ctx.L.cwslg_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(8 * n))
assert rc == 0
st = buf.reshape(n, 8).astype(np.int64)
st = st[:, [0, 7, 1, 2, 3, 4, 5, 6]]          # chronological order: entry, decoded, loads issued, ...
d = np.diff(st, axis=1)
ok = (d > 0).all(axis=1) & (d < 10 ** 7).all(axis=1)
d = d[ok]
names = ["taps+decode (persist: 0)", "issue loads (persist: 0, prefetch is inside phase1 span)", "phase0+barrierA", "wait vmcnt(0)", "phase1+barrierB", "phase2+barrierC", "epilogue"]
tot = (st[ok, 7] - st[ok, 0])
print("workgroups", len(d), " s_memtime ticks per tile (median / mean):")
for k, nme in enumerate(names):
    print("  %-18s %8.0f %8.0f  %5.1f%%" % (nme, np.median(d[:, k]), d[:, k].mean(), 100 * d[:, k].mean() / tot.mean()))
print("  %-18s %8.0f %8.0f" % ("total", np.median(tot), tot.mean()))
