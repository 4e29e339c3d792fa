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
    # a retune still in force at a boundary does not outlive it: Instance re-creates its SSBD from its own demodFreq (Instance.cpp:251)
    ctx.channel_tune(ch, 60000); oc.tune(60000)
    ctx.push_iq(rx, iq[:16 * blk]); oc.push_many(iq[:16 * blk])
    ctx.slot_boundary("FT8", 35); oc.boundary(35)
    ctx.push_iq(rx, iq[:16 * blk]); oc.push_many(iq[:16 * blk])
    ctx.slot_boundary("FT8", 50)
    ref = oc.boundary(50, want_f32=True)
    a, _ = ctx.fetch_audio_f32(ch)
    if exact:
        assert np.array_equal(a.view(np.uint32), ref["f32"].view(np.uint32))
    else:
        assert float(np.abs(a.astype(np.float64) - ref["f32"]).max()) <= 1e-5 * float(np.abs(ref["f32"]).max())
    assert float(np.abs(a[200:16 * blk // 16]).max()) > 1000.0     # the -26 kHz tone again, not the 60 kHz one


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("fs,blk", [(192000, 2048), (96000, 1024), (48000, 64)])
def test_retune_without_reset_matches_oracle(ctx, oracle, exact, fs, blk):
    """cwslg_channel_tune_ex(reset = 0) == SSBD::Tune(F, isUSB, false) (SSBD.hpp:97, :116-121 skipped): filter history, block
    position and phase survive; the 31 outputs after the retune blend both tunings.  Against the oracle, which
    tests/test_oracle_vs_ref.py pins to the compiled header for exactly this call."""
    D = fs // 12000
    ctx.set_exact(exact)
    rx = ctx.receiver_open(fs, blk, 0)
    f0, f1, f2 = -fs // 8, fs // 4, 1234
    ch = ctx.channel_open(rx, f0, "FT8")
    oc = oracle.Channel("FT8", fs, blk, f0)
    n_blk = max(24, 6 * 2048 // blk)
    iq = oracle.synth_iq(11, n_blk * blk, fs, tones_hz=[f0 + 900.0, f1 + 1500.0, f2 - 800.0], amp=1.0e4)
    oc.boundary(5); ctx.slot_boundary("FT8", 5)
    cuts = [0, (n_blk // 3) * blk, (2 * n_blk // 3) * blk, n_blk * blk]
    plan = [None, (f1, True), (f2, False)]                      # the second retune also flips the sideband
    for k in range(3):
        if plan[k] is not None:
            ctx.channel_tune(ch, plan[k][0], plan[k][1], reset=False); oc.tune(plan[k][0], plan[k][1], reset=False)
        piece = iq[cuts[k]:cuts[k + 1]]
        if k == 1 and blk <= 256:                               # the transition split over several launches
            for q in range(0, len(piece), blk):
                ctx.push_iq(rx, piece[q:q + blk]); ctx.process()
        else:
            ctx.push_iq(rx, piece)
        oc.push_many(piece)
    ctx.slot_boundary("FT8", 20)
    ref = oc.boundary(20, want_f32=True)
    a, nv = ctx.fetch_audio_f32(ch)
    g = ctx.fetch_frame(ch)
    assert nv == n_blk * blk // D
    if exact:
        bad = np.nonzero(a.view(np.uint32) != ref["f32"].view(np.uint32))[0]
        assert bad.size == 0, (bad[:8], cuts[1] // D, cuts[2] // D)
        assert np.array_equal(g["i16"], ref["i16"])
    else:
        peak = float(np.abs(ref["f32"]).max())
        assert float(np.abs(a.astype(np.float64) - ref["f32"]).max()) <= 1e-5 * peak
        assert int(np.abs(g["i16"].astype(np.int32) - ref["i16"]).max()) <= 1
    # not the reset form: the outputs right after the first retune differ from a demodulator restarted there
    o2 = oracle.Channel("FT8", fs, blk, f0)
    o2.boundary(5); o2.push_many(iq[:cuts[1]]); o2.tune(f1, True, reset=True); o2.push_many(iq[cuts[1]:])
    r2 = o2.boundary(20, want_f32=True)
    k1 = cuts[1] // D
    assert not np.array_equal(r2["f32"][k1:k1 + 31], ref["f32"][k1:k1 + 31])
    # the frame after the next boundary comes from a NEW demodulator built from the Instance's own frequency (Instance.cpp:251):
    # the retunes are gone
    more = oracle.synth_iq(12, 8 * blk if blk >= 1024 else 64 * blk, fs, tones_hz=[f2 - 800.0, f0 + 900.0], amp=1.0e4)
    ctx.push_iq(rx, more); oc.push_many(more)
    ctx.slot_boundary("FT8", 35)
    ref = oc.boundary(35, want_f32=True)
    a, _ = ctx.fetch_audio_f32(ch)
    if exact:
        assert np.array_equal(a.view(np.uint32), ref["f32"].view(np.uint32))
    else:
        assert float(np.abs(a.astype(np.float64) - ref["f32"]).max()) <= 1e-5 * float(np.abs(ref["f32"]).max())


def test_second_retune_inside_the_transition_is_refused(ctx):
    rx = ctx.receiver_open(192000, 64, 0)
    ch = ctx.channel_open(rx, 1000, "FT8")
    ctx.slot_boundary("FT8", 5)
    iq = (np.ones(64 * 40) + 0j).astype(np.complex64)
    ctx.push_iq(rx, iq)
    ctx.channel_tune(ch, 2000, reset=False)
    ctx.push_iq(rx, iq[:64 * 4])                               # 16 blocks: the transition (32) is still open
    with pytest.raises(P.CwslGpuError) as e:
        ctx.channel_tune(ch, 3000, reset=False)
    assert e.value.status == -10                                # CWSLG_ERR_UNSUPPORTED
    ctx.push_iq(rx, iq[:64 * 8])                               # past it
    ctx.channel_tune(ch, 3000, reset=False)
    ctx.channel_tune(ch, 4000)                                  # and the reset form is always allowed
