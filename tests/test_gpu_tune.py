"""GPU: cwslg_channel_tune == SSBD::Tune(F, isUSB) on a live channel (SSBD.hpp:96-123)."""
import numpy as np
import pytest

import cwsl_digi_amd as P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("exact", [True, False])
def test_retune_mid_slot_matches_oracle(ctx, oracle, exact):
    fs, blk = 192000, 2048
    ctx.set_exact(exact)
    rx = ctx.receiver_open(fs, blk, 0)
    ch = ctx.channel_open(rx, -26000, "FT8")
    other = ctx.channel_open(rx, 41000, "FT8")                 # an untouched neighbour on the same receiver
    oc, oo = oracle.Channel("FT8", fs, blk, -26000), oracle.Channel("FT8", fs, blk, 41000)
    iq = oracle.synth_iq(8, 120 * blk, fs, tones_hz=[-26000 + 900.0, 60000 + 1500.0, 41000 + 700.0], amp=1.0e4)
    for c_ in (oc, oo):
        c_.boundary(5)
    ctx.slot_boundary("FT8", 5)
    cuts = [0, 37 * blk, 81 * blk, 120 * blk]
    tunes = [None, 60000, -26000]                               # retune before the 2nd and the 3rd piece
    for k in range(3):
        if tunes[k] is not None:
            ctx.channel_tune(ch, tunes[k]); oc.tune(tunes[k])
        piece = iq[cuts[k]:cuts[k + 1]]
        ctx.push_iq(rx, piece)
        oc.push_many(piece); oo.push_many(piece)
    # a rejected retune leaves the channel as it was (the reference throws before storing anything)
    with pytest.raises(P.CwslGpuError) as e:
        ctx.channel_tune(ch, 95000)
    assert e.value.status == -3 and "high" in str(e.value)
    with pytest.raises(P.CwslGpuError) as e:
        ctx.channel_tune(ch, -97000)
    assert e.value.status == -2
    ctx.slot_boundary("FT8", 20)
    for gch, och in ((ch, oc), (other, oo)):
        ref = och.boundary(20, want_f32=True)
        a, nv = ctx.fetch_audio_f32(gch)
        g = ctx.fetch_frame(gch)
        assert nv == 120 * blk // 16
        if exact:
            assert np.array_equal(a.view(np.uint32), ref["f32"].view(np.uint32)) and np.array_equal(g["i16"], ref["i16"])
        else:
            peak = float(np.abs(ref["f32"]).max())
            assert float(np.abs(a.astype(np.float64) - ref["f32"]).max()) <= 1e-5 * peak
            assert int(np.abs(g["i16"].astype(np.int32) - ref["i16"]).max()) <= 1
    # the retuned stretch really carries the other signal: energy appears where the 60 kHz tone was selected
    a, _ = ctx.fetch_audio_f32(ch)
    seg = a[37 * blk // 16 + 200: 81 * blk // 16]
    assert float(np.abs(seg).max()) > 1000.0
