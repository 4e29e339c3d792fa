"""Diagnostic: where a workgroup of ft8_sync2d_v2_kernel spends its time (s_memtime stamps of a -DCWSLG_STAMP -DCWSLG_STAMP_SYNC
lab build, wave 0: staging, then the phases of its FIRST bin, then the end of its four bins)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CWSLG_LIB"] = os.path.join(ROOT, os.environ.get("CWSLG_STAMP_LIB", "cwsl_digi_amd/lib/libcwslgpu_stamp.so"))
os.environ.setdefault("CWSLG_SYNC_VARIANT", "0")
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
st = buf.reshape(n, 8).astype(np.int64)[:14848, :7]      # 29 bands x 512 channels
d = np.diff(st, axis=1)
ok = (d > 0).all(axis=1) & (d < 10 ** 7).all(axis=1)
d = d[ok]
names = ["staging (loads, LDS scatter, barrier)", "bin 1: 7-tone sums (c0) + wave barrier", "bin 1: 42 reads + Costas sums", "bin 1: 2 x sync_finish (8 divisions)",
         "bin 1: two peak searches + store", "bins 2..4"]
tot = st[ok, 6] - st[ok, 0]
print("workgroups", len(d), " s_memtime ticks (100 MHz? see total vs kernel time), wave 0 (median / mean / share):")
for k, nme in enumerate(names):
    print("  %-44s %8.0f %8.0f  %5.1f%%" % (nme, np.median(d[:, k]), d[:, k].mean(), 100 * d[:, k].mean() / tot.mean()))
print("  %-44s %8.0f %8.0f" % ("total", np.median(tot), tot.mean()))
span = st[ok, 6].max() - st[ok, 0].min()
print("first entry -> last exit of the stamped workgroups (ticks):", int(span))
