"""Helper of tests/test_gpu_demod.py (not a test): one slot on four channels through whatever library and kernel variant the
environment selects (CWSLG_LIB=lab, CWSLG_DEMOD_VARIANT=n), checked against the oracle.  argv[1] = "fast" | "exact"."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import cwsl_digi_amd as P                      # noqa: E402
from oracle import oracle as O                 # noqa: E402

FS, IQ_LEN = 192000, 2048
FREQS = [0, 1234, 24000, 87000]
exact = sys.argv[1] == "exact"
na, nb = 24 * IQ_LEN, 90 * IQ_LEN
tones = sum(([f + 700.0, f + 1500.5, f + 2600.25] for f in FREQS), [])
iq = O.synth_iq(0xBEEF, na + nb, FS, tones_hz=tones, amp=2.0e4)
with P.Context(0) as ctx:
    ctx.set_exact(exact)
    rx = ctx.receiver_open(FS, IQ_LEN, 0)
    chans = [ctx.channel_open(rx, f, "FT8") for f in FREQS]
    ctx.push_iq(rx, iq[:na]); ctx.slot_boundary("FT8", 15)
    ctx.push_iq(rx, iq[na:]); ctx.slot_boundary("FT8", 30)
    name = ctx.demod_kernel_name()
    for f, ch in zip(FREQS, chans):
        oc = O.Channel("FT8", FS, IQ_LEN, f)
        oc.push_many(iq[:na]); assert oc.boundary(15) is None
        oc.push_many(iq[na:]); ref = oc.boundary(30, want_f32=True)
        f32, nv = ctx.fetch_audio_f32(ch)
        assert nv == nb // 16
        if exact:
            assert np.array_equal(f32.view(np.uint32), ref["f32"].view(np.uint32)), f"{name}: bits differ"
            assert np.array_equal(ctx.fetch_frame(ch)["i16"], ref["i16"])
        else:
            peak = float(np.abs(ref["f32"]).max())
            err = float(np.abs(f32.astype(np.float64) - ref["f32"]).max())
            assert err <= 1e-5 * peak, (name, err, peak)
print("lab check OK:", name)
