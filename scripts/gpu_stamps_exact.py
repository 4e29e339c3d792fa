"""Diagnostic: where a tile of demod_exact3_kernel spends its cycles (s_memtime stamps of a -DCWSLG_STAMP build, wave 0 of each workgroup)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CWSLG_LIB"] = os.path.join(ROOT, os.environ.get("CWSLG_STAMP_LIB", "cwsl_digi_amd/lib/libcwslgpu_stamp.so"))
import cwsl_digi_amd as P
ctx = P.Context(0)
ctx.set_exact(True)
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
raw = buf.reshape(n, 8)
used = raw[:, 7] != 0
hw = raw[used, 6]
t_start = raw[used, 7].astype(np.int64)
xcc = (hw >> 32) & 0xF
cu = (hw & 0xFFFFFFFF) >> 8 & 0xFF          # cu_id[11:8], sh_id[12], se_id[15:13]
key = xcc * 256 + cu
vals, cnt = np.unique(key, return_counts=True)
print("workgroups started:", int(used.sum()), " distinct (xcc, se/sh/cu):", len(vals), " workgroups per CU: min %d max %d" % (cnt.min(), cnt.max()),
      " histogram", dict(zip(*np.unique(cnt, return_counts=True))))
lo = hw & 0xFFFFFFFF
wid, simd, tg = lo & 0xF, (lo >> 4) & 3, (lo >> 16) & 0xF
for nm, fld in (("wave_id", wid), ("simd_id", simd), ("tg_id", tg)):
    per_cu = {}
    for k_, f_ in zip(key.tolist(), fld.tolist()):
        per_cu.setdefault(k_, []).append(int(f_))
    pats = {}
    for v in per_cu.values():
        pats[tuple(sorted(v))] = pats.get(tuple(sorted(v)), 0) + 1
    print("HW_ID.%s of wave 0 of the workgroups sharing a CU:" % nm, dict(sorted(pats.items(), key=lambda kv: -kv[1])[:6]))
print("start spread (ticks): p50 %d p90 %d max %d" % tuple(np.percentile(t_start - t_start.min(), [50, 90, 100]).astype(int)))
st = raw.astype(np.int64)[:, :6]
el = (st[used, 5] - t_start)
print("entry -> end of the stamped tile (ticks): median %d  => per tile over 101 tiles: %.0f" % (np.median(el), np.median(el) / 101.0))
d = np.diff(st, axis=1)
ok = (d > 0).all(axis=1) & (d < 10 ** 7).all(axis=1)
d = d[ok]
names = ["decode + issue loads", "phasor rebuild (incl. ckpt wait)", "wait IQ + mix + barrier", "FIR (33 steps)", "store + peak"]
tot = (st[ok, 5] - st[ok, 0])
print("workgroups", len(d), " s_memtime ticks per tile, wave 0 (median / mean):")
for k, nme in enumerate(names):
    print("  %-34s %8.0f %8.0f  %5.1f%%" % (nme, np.median(d[:, k]), d[:, k].mean(), 100 * d[:, k].mean() / tot.mean()))
print("  %-34s %8.0f %8.0f" % ("total", np.median(tot), tot.mean()))
fir = d[:, 3]
print("FIR ticks percentiles 5/25/50/75/95:", np.percentile(fir, [5, 25, 50, 75, 95]).astype(int))
