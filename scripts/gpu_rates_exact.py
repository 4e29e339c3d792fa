"""Exact-mode demod launch time at 48 / 96 / 192 kHz (512 FT8 slots, no sync stage): the product's assembly FIR against the C++ form
(lab library, CWSLG_DEMOD_VARIANT=25).  Usage: python scripts/gpu_rates_exact.py  (set CWSLG_LIB / CWSLG_DEMOD_VARIANT to pick the form)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import cwsl_digi_amd as P
for fs in (48000, 96000, 192000):
    ctx = P.Context(0)
    ctx.set_exact(True)
    ctx.set_timing(True)
    S, BLK = 512, 2048
    N = 15 * fs
    rb = N // BLK + 3
    for s in range(S):
        rx = ctx.receiver_open(fs, BLK, 0, ring_blocks=rb)
        cap = rb * BLK
        ctx.push_synth(rx, s, cap // 2, BLK); ctx.push_synth(rx, s, cap - cap // 2, BLK)
        ctx.channel_open(rx, -fs // 2 + 3000 + (s * 4373) % (fs - 12000), "FT8")
    ctx.slot_boundary("FT8", 1)
    for k in range(2):
        ctx.ring_commit_all(N // BLK * BLK, BLK); ctx.process(); ctx.slot_boundary("FT8", 2 + k)
    ctx.synchronize(); ctx.reset_stats()
    K = 8
    for k in range(K):
        ctx.ring_commit_all(N // BLK * BLK, BLK); ctx.process(); ctx.slot_boundary("FT8", 4 + k)
    ctx.synchronize()
    st = ctx.stats()
    ms = st["demod_ms"] / max(1, st["demod_launches"])
    print("fs %6d: %s  %.3f ms per launch = %.1f G samples/s" % (fs, ctx.demod_kernel_name(), ms, S * (N // BLK * BLK) / ms / 1e6))
    ctx.close()
