"""How far the default (fused) mode is from the reference arithmetic when the frame carries little in-band energy next to a strong
out-of-band signal: the 1e-5 bound is relative to the FRAME peak, the rounding noise to the input level.  Exact mode is the remedy."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import cwsl_digi_amd as P
from oracle import oracle as O
fs, blk = 192000, 2048
for amp_in, amp_out in [(0.0, 2.0e4), (20.0, 2.0e4), (200.0, 2.0e4), (2.0e4, 2.0e4)]:
    for exact in (False, True):
        with P.Context(0) as ctx:
            ctx.set_exact(exact)
            rx = ctx.receiver_open(fs, blk, 0)
            ch = ctx.channel_open(rx, -26000, "FT8")
            oc = O.Channel("FT8", fs, blk, -26000)
            n = 200 * blk
            t = np.arange(n) / fs
            rng = np.random.default_rng(1)
            iq = (amp_out * np.exp(2j * np.pi * 41000.0 * t) + amp_in * np.exp(2j * np.pi * (-26000 + 1200.0) * t)
                  + (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.5).astype(np.complex64)
            ctx.slot_boundary("FT8", 5); oc.boundary(5)
            ctx.push_iq(rx, iq); oc.push_many(iq)
            ctx.slot_boundary("FT8", 20)
            ref = oc.boundary(20, want_f32=True)
            a, nv = ctx.fetch_audio_f32(ch)
            peak = float(np.abs(ref["f32"]).max())
            err = float(np.abs(a.astype(np.float64) - ref["f32"]).max())
            g = ctx.fetch_frame(ch)
            print("in-band amp %8.1f, out-of-band amp %8.1f, exact=%d: frame peak %.4g, max err %.3g = %.3g of peak, int16 max diff %d"
                  % (amp_in, amp_out, exact, peak, err, err / peak, int(np.abs(g["i16"].astype(np.int32) - ref["i16"]).max())))
