"""Demod launch time at 48 / 96 / 192 kHz in both arithmetic modes (512 FT8 slots unless --slots N, no sync stage), with the in-kernel clock: one JSON line.
Usage: python scripts/gpu_rates_exact.py [--slots N] (set CWSLG_LIB / CWSLG_DEMOD_VARIANT to pick a measured alternative from the lab library)."""
import json, os, sys
SLOTS = int(sys.argv[sys.argv.index("--slots") + 1]) if "--slots" in sys.argv else 512
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import cwsl_digi_amd as P
out = []
for fs in (48000, 96000, 192000):
    for exact in (True, False):
        ctx = P.Context(0)
        ctx.set_exact(exact)
        S, BLK = SLOTS, 2048
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
        ctx.synchronize(); ctx.reset_stats(); ctx.set_timing(True)
        K = 8
        for k in range(K):
            ctx.ring_commit_all(N // BLK * BLK, BLK); ctx.process(); ctx.slot_boundary("FT8", 4 + k)
        ctx.synchronize()
        st = ctx.stats()
        ms = st["demod_ms"] / max(1, st["demod_launches"])
        rec = {"fs_hz": fs, "mode": "exact" if exact else "fast", "kernel": ctx.demod_kernel_name(), "slots": S, "ms_per_launch": ms,
               "g_samples_per_s": S * (N // BLK * BLK) / ms / 1e6, "hbm_frac": (8.0 + 4.0 * 12000 / fs) * S * (N // BLK * BLK) / (ms * 1e-3) / 8e12,
               "clock_mhz": st["demod_clock_mhz"]}
        out.append(rec)
        print("fs %6d %-5s: %-34s %.3f ms per launch = %6.1f G samples/s = %.3f of 8 TB/s, clock %.0f MHz" % (fs, rec["mode"], rec["kernel"], ms, rec["g_samples_per_s"], rec["hbm_frac"], rec["clock_mhz"]), file=sys.stderr)
        ctx.close()
print(json.dumps({"rates": out}))
