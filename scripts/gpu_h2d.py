"""PCIe-inclusive rate: host buffers through cwslg_push_iq (pageable numpy -> pinned staging -> H2D -> demod)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cwsl_digi_amd as P
ctx = P.Context(0)
S, BLK, NB = 64, 2048, 940            # 64 receivers x 10 s
rxs = [ctx.receiver_open(192000, BLK, 0) for _ in range(S)]
for k, rx in enumerate(rxs):
    ctx.channel_open(rx, -80000 + 2500 * k, "FT8")
ctx.slot_boundary("FT8", 1)
rng = np.random.default_rng(0)
blk = (rng.standard_normal(BLK * 64) + 1j * rng.standard_normal(BLK * 64)).astype(np.complex64) * 1000
ctx.push_iq(rxs[0], blk); ctx.synchronize()
t0 = time.perf_counter()
for it in range(NB // 64):
    for rx in rxs:
        ctx.push_iq(rx, blk)            # 64 blocks per call
ctx.slot_boundary("FT8", 2)
ctx.synchronize()
dt = time.perf_counter() - t0
n = S * (NB // 64) * 64 * BLK
print("host-buffer path: %.1f M samples/s = %.2f GB/s over PCIe, %.0f real-time 192 kHz streams" % (n / dt / 1e6, n * 8 / dt / 1e9, n / dt / 192000))
