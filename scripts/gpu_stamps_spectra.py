"""Diagnostic: where a transform of symbol_spectra_v2_kernel spends its time (s_memtime stamps of a -DCWSLG_STAMP -DCWSLG_STAMP_SPEC lab build;
every wave of every workgroup, its middle transform)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CWSLG_LIB"] = os.path.join(ROOT, os.environ.get("CWSLG_STAMP_LIB", "cwsl_digi_amd/lib/libcwslgpu_stamp.so"))
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
rc = ctx.L.cwslg_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(8 * n))
assert rc == 0
nwg = 31 * S
st = buf.reshape(n // 4, 4, 8).astype(np.int64)[:min(nwg, 16384)]
names = ["convert + issue prefetch + stage 1", "barrier 1", "pass A", "pass B", "barrier 2", "last stage + unpack", "barrier 3"]
for wv, label in ((0, "wave 0 (stage 1: k2 = 0, 1, 4)"), (1, "wave 1 (same)"), (2, "wave 2 (stage 1: k2 = 2, 3)"), (3, "wave 3 (same; + the bins above 960)")):
    s_ = st[:, wv, :]
    d = np.diff(s_, axis=1)
    ok = (d > 0).all(axis=1) & (d < 10 ** 6).all(axis=1)
    d = d[ok]
    tot = (s_[ok, 7] - s_[ok, 0])
    print(label, ": workgroups", len(d), " ticks of one transform (median / mean / share):")
    for k, nme in enumerate(names):
        print("  %-40s %8.0f %8.0f  %5.1f%%" % (nme, np.median(d[:, k]), d[:, k].mean(), 100 * d[:, k].mean() / tot.mean()))
    print("  %-40s %8.0f %8.0f" % ("total", np.median(tot), tot.mean()))
