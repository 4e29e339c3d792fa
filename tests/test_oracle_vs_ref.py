"""CPU: oracle restatement vs the compiled reference headers (oracle/_ref), where that build exists
(the build container; on the GPU box the prebuilt .so travels with the snapshot)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref/libcwsl_ref.so not built (needs /root/reference)")
    return oracle


@pytest.mark.parametrize("fs,f,usb", [(192000, -26000, True), (192000, 90000, True), (192000, 12000, False),
                                      (96000, -30000, True), (48000, 9000, True), (48000, -3000, False)])
def test_bit_identical_audio_and_phasor(ref, fs, f, usb):
    D = fs // 12000
    n = 4 * D * 3000
    iq = ref.synth_iq(fs ^ (f & 0xFFFF), n, fs, tones_hz=[f + 1000.0, f - 1800.0 if not usb else f + 2222.0], amp=1e4)
    a, ta = ref.Demod(fs, float(f), usb=usb).run(iq, trace=True)
    b, tb = ref.RefDemod(fs, float(f), usb=usb).run(iq, trace=True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(ta.view(np.uint64), tb.view(np.uint64))


def test_getters_and_errors(ref):
    r = ref.RefDemod(192000, 0.0)
    R = ref.ref()
    assert (R.ref_ssbd_in_size(r.h), R.ref_ssbd_out_size(r.h), R.ref_ssbd_out_rate(r.h), R.ref_ssbd_delay(r.h)) == (64, 4, 12000, 8)
    for args, msg in (((192000, 97000.0), "low"), ((192000, 93000.0), "high")):
        with pytest.raises(ValueError) as e1:
            ref.RefDemod(*args)
        with pytest.raises(ValueError) as e2:
            ref.Demod(*args)
        assert msg in str(e1.value) and str(e1.value) == str(e2.value)
    with pytest.raises(ValueError) as e1:
        ref.RefDemod(44100, 0.0)
    with pytest.raises(ValueError) as e2:
        ref.Demod(44100, 0.0)
    assert str(e1.value) == str(e2.value) == "Fs/B must be an even integer >= 4"


def test_lowpass_design_matches(ref):
    for order, bw in ((512, 6000 / 192000.), (256, 6000 / 96000.), (128, 0.125)):
        a = np.empty(order, np.float32); b = np.empty(order, np.float32)
        ref.lib().orc_lowpass_design(order, bw, a)
        ref.ref().ref_build_lowpass(order, bw, b)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


# ---- slot framing: the oracle's Channel against Instance's sequences replayed on the reference's own containers ----
@pytest.mark.parametrize("mode,fs,blk,f", [("FT4", 48000, 1024, 3000), ("FT8", 96000, 2048, -20000), ("FT4", 192000, 2048, 50000)])
def test_framing_matches_reference_containers(ref, mode, fs, blk, f):
    """Random schedules of blocks and boundaries, including over-full slots ("af buffer full"), empty slots and the
    discarded first frame: same drop decisions, same epochs, same sample counts, bit-identical float frames."""
    rng = np.random.default_rng(fs + blk)
    oc = ref.Channel(mode, fs, blk, f)
    ri = ref.RefInstance(mode, fs, blk, f)
    dec = fs // 12000
    cap_blocks = (oc.frame_len - 1) // (blk // dec)              # blocks a frame accepts before the guard trips
    epoch = 1000
    n_emit = n_drop = 0
    for slot in range(6):
        n_blocks = [cap_blocks // 3, 0, cap_blocks + 5, 1, cap_blocks, 7][slot]
        iq = ref.synth_iq(100 + slot, max(1, n_blocks) * blk, fs, tones_hz=[f + 1500.0], amp=8000.0)
        for k in range(n_blocks):
            b = iq[k * blk:(k + 1) * blk]
            took_o, took_r = oc.push(b), ri.push(b)
            assert took_o == took_r
            n_drop += 1 - took_r
        epoch += int(rng.integers(1, 20))
        fo = oc.boundary(epoch, want_f32=True)
        emitted, t0, nw, fr = ri.boundary(epoch)
        assert (fo is not None) == emitted
        if emitted:
            n_emit += 1
            assert fo["t_start"] == t0
            assert np.array_equal(fo["f32"].view(np.uint32), fr.view(np.uint32))
    assert n_emit == 5 and n_drop >= 5                           # first frame discarded; the over-full slot dropped blocks


def test_mode_groups_match_syncpredicates(ref):
    import cwsl_digi_amd as P
    modes = ["FT8", "JS8", "FT4", "WSPR", "Q65-30", "JT65", "FST4-60", "FST4-120", "FST4-300", "FST4-900", "FST4-1800",
             "FST4W-120", "FST4W-300", "FST4W-900", "FST4W-1800"]
    for m in modes:
        g = ref.ref_mode_group(m)
        assert g == P.group_of(m) == P.parse_decoder_line(f"14074000 {m}", 1.0)["group"], m
    assert ref.ref_mode_group("PSK31") == -1
    with pytest.raises(Exception):
        P.parse_decoder_line("14074000 PSK31", 1.0)


def test_locator_and_trim_helpers_match_reference(ref):
    import random
    import cwsl_digi_amd as P
    rng = random.Random(5)
    alphabet = "ABfn0199 -/x"
    for _ in range(400):
        loc = "".join(rng.choice(alphabet) for _ in range(rng.choice([3, 4, 4, 4, 5])))
        if " " in loc:
            continue
        line = f"123045 -12  0.3 1234 ~  CQ K1ABC {loc}"
        got = P.parse_decode_line("FT8", line, 0)
        want = ref.parse_decode_line("FT8", line, 0)
        assert got == want
        if got["status"] == "ok" and got["call"] == "K1ABC":
            assert (got["locator"] is not None) == ref.ref_is_valid_locator(loc), loc
    for _ in range(200):
        pad_l = "".join(rng.choice(" \t") for _ in range(rng.randrange(0, 4)))
        pad_r = "".join(rng.choice(" \t\r\n") for _ in range(rng.randrange(0, 4)))
        core = "123045 -12  0.3 1234 ~  CQ K1ABC FN42"
        assert ref.ref_trim(pad_l + core + pad_r) == core
        assert P.parse_decode_line("FT8", pad_l + core + pad_r, 0)["call"] == "K1ABC"


def test_tune_on_a_live_demodulator_matches_reference(ref):
    """SSBD::Tune(F, isUSB) mid-stream: the oracle's orc_demod_tune against the compiled header, incl. a rejected retune."""
    fs = 96000
    iq = ref.synth_iq(3, 64 * 32 * 12, fs, tones_hz=[-20000 + 1000.0, 31000 + 1700.0], amp=9000.0)
    a, b = ref.Demod(fs, -20000), ref.RefDemod(fs, -20000)
    n1 = 64 * 32 * 5
    ya, yb = a.run(iq[:n1]), b.run(iq[:n1])
    assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32))
    for obj in (a, b):
        with pytest.raises(ValueError, match=r"Signal outside of band \(high\)"):
            obj.tune(45000)                                   # |F + B| > Fs/2: thrown before anything is stored
    a.tune(31000); b.tune(31000)
    ya, yb = a.run(iq[n1:]), b.run(iq[n1:])
    assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32))
    assert np.array_equal(a.tone.view(np.uint32), b.tone.view(np.uint32))
    # a retuned object equals a freshly constructed one from that point on (workspace, index and phase are reset)
    c = ref.Demod(fs, 31000)
    assert np.array_equal(c.run(iq[n1:]).view(np.uint32), ya.view(np.uint32))


def test_tune_without_reset_matches_reference(ref):
    """SSBD::Tune(F, isUSB, reset = false) (SSBD.hpp:97,116-121): workspace, index and phase survive the retune.  The oracle's
    orc_demod_tune_ex against the compiled header, bit for bit, through two retunes (one of them flipping the sideband)."""
    fs = 192000
    iq = ref.synth_iq(5, 64 * 40 * 9, fs, tones_hz=[-26000 + 900.0, 60000 + 1500.0, 1234 + 700.0], amp=9000.0)
    a, b = ref.Demod(fs, -26000), ref.RefDemod(fs, -26000)
    cuts = [0, 64 * 40 * 3, 64 * 40 * 6, 64 * 40 * 9]
    plan = [None, (60000, True), (1234, False)]
    outs = []
    for k in range(3):
        if plan[k] is not None:
            a.tune(plan[k][0], plan[k][1], reset=False); b.tune(plan[k][0], plan[k][1], reset=False)
        ya, yb = a.run(iq[cuts[k]:cuts[k + 1]]), b.run(iq[cuts[k]:cuts[k + 1]])
        assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32)), k
        outs.append(ya)
    # and it is NOT the reset form: the 31 outputs after the retune still carry the old tuning's partial sums
    c = ref.Demod(fs, -26000)
    c.run(iq[:cuts[1]]); c.tune(60000, True, reset=True)
    yc = c.run(iq[cuts[1]:cuts[2]])
    assert not np.array_equal(yc[:31].view(np.uint32), outs[1][:31].view(np.uint32))
