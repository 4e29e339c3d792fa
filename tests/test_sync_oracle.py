"""CPU: the FT8 sync restatement (oracle/sync_oracle.c).  PARITY UNPINNED by the reference -- these tests pin
the restatement to independent numerics (numpy FFT) and to the algorithm's defining behaviour (it finds Costas
arrays where they are)."""
import numpy as np
import pytest

from ft8_signal import ft8_iq


def _frame_with_signals(oracle, specs, seed=5, noise=True):
    fs, blk, f = 192000, 2048, 10000
    n = 2880000 // blk * blk
    rng = np.random.default_rng(seed)
    iq = oracle.synth_iq(seed, n, fs) if noise else np.zeros(n, np.complex64)
    for audio_hz, t0, amp in specs:
        iq = iq + ft8_iq(fs, n, f, audio_hz, t0, amp, rng)
    c = oracle.Channel("FT8", fs, blk, f)
    c.boundary(1)
    c.push_many(iq.astype(np.complex64))
    return c.boundary(2)["i16"]


def test_spectra_match_numpy_fft(oracle):
    fr = _frame_with_signals(oracle, [(1000.0, 0.5, 3000.0)])
    s = oracle.ft8_spectra(fr, 1024)
    for j in (0, 57, 371):
        x = np.zeros(3840); x[:1920] = fr[480 * j:480 * j + 1920].astype(np.float32) * np.float32(1 / 300.0)
        ref = np.abs(np.fft.rfft(x)) ** 2
        assert np.abs(s[j] - ref[:1024]).max() <= 2e-6 * ref.max()


def test_finds_costas_arrays_at_the_right_bin_and_lag(oracle):
    specs = [(700.0, 0.5, 3000.0), (1531.25, 1.3, 2000.0), (2400.0, 0.1, 1500.0)]
    fr = _frame_with_signals(oracle, specs)
    cands = oracle.ft8_sync(fr, 200, 3000, 1.5, 200)
    assert len(cands) >= 3
    syncs = [c[2] for c in cands]
    assert syncs == sorted(syncs, reverse=True)
    for audio_hz, t0, _ in specs:
        want_bin = int(round(audio_hz / 3.125))
        want_lag = (t0 - 0.44) / 0.04            # jstrt = int(0.5/0.04) = 12 steps, 1-based step index
        hit = [c for c in cands if abs(c[0] - want_bin) <= 1 and abs(c[1] - want_lag) <= 1.0]
        assert hit, (audio_hz, t0, cands[:6])
        assert hit[0][2] > 3.0
    # the three injected signals outrank everything else
    top = {(int(round(c[3] / 50.0))) for c in cands[:3]}
    assert top == {int(round(a / 50.0)) for a, _, _ in specs}


def test_noise_only_frame_has_few_weak_candidates(oracle):
    fr = _frame_with_signals(oracle, [])
    cands = oracle.ft8_sync(fr, 200, 3000, 2.5, 200)
    assert all(c[2] < 6.0 for c in cands)


def test_candidate_fields_and_dedupe(oracle):
    fr = _frame_with_signals(oracle, [(1000.0, 0.5, 4000.0)])
    cands = oracle.ft8_sync(fr, 200, 3000, 1.5, 600)
    for b, lag, sy, fhz, dt in cands:
        assert fhz == np.float32(b) * np.float32(3.125) and dt == (np.float32(lag) - np.float32(0.5)) * np.float32(0.04)
        assert -62 <= lag <= 62 and 64 <= b <= 960
    # near-duplicates (|df|<4 Hz and |dt|<0.04 s) never both survive
    for i, a in enumerate(cands):
        for b in cands[:i]:
            fd = abs(np.float32(a[3]) - np.float32(b[3])); td = abs(np.float32(a[4]) - np.float32(b[4]))   # float32, as the code
            assert not (fd < np.float32(4.0) and td < np.float32(0.04))


# ---------------------------------------------------------------------------------------------- FT4
from ft8_signal import ft4_iq


def _ft4_frame(oracle, specs, seed=9):
    fs, blk, f = 192000, 2048, -40000
    n = 1440000 // blk * blk
    rng = np.random.default_rng(seed)
    iq = oracle.synth_iq(seed, n, fs)
    for audio_hz, t0, amp in specs:
        iq = iq + ft4_iq(fs, n, f, audio_hz, t0, amp, rng)
    c = oracle.Channel("FT4", fs, blk, f)
    c.boundary(1)
    c.push_many(iq.astype(np.complex64))
    return c.boundary(2)["i16"]


def test_ft4_spectra_match_numpy_fft(oracle):
    fr = _ft4_frame(oracle, [(1000.0, 0.5, 3000.0)])
    s = oracle.ft4_spectra(fr)
    i = np.arange(2304); pi = np.pi
    win = (0.3635819 - 0.4891775 * np.cos(2 * pi * i / 2304) + 0.1365995 * np.cos(4 * pi * i / 2304)
           - 0.0106411 * np.cos(6 * pi * i / 2304)).astype(np.float32)
    for j in (0, 60, 121):
        x = fr[576 * j:576 * j + 2304].astype(np.float32) * np.float32(1 / 300.0) * win
        ref = np.abs(np.fft.rfft(x.astype(np.float64))) ** 2
        assert np.abs(s[j] - ref).max() <= 2e-6 * ref.max()


def test_fixed_log_exp_are_accurate(oracle):
    import math
    L = oracle.lib()
    for x in (1e-12, 3e-5, 0.3, 1.0, 2.5, 1234.5, 7e12):
        assert abs(L.orc_log10_fixed(x) - math.log10(x)) <= 4e-15 * max(1.0, abs(math.log10(x)))
    for y in (-6.0, -3.2, -0.5, 0.0, 0.33, 2.75, 9.1):
        assert abs(L.orc_exp10_fixed(y) / 10 ** y - 1) <= 4e-15


def test_ft4_finds_signals(oracle):
    specs = [(700.0, 0.3, 2500.0), (1800.0, 0.6, 1500.0), (3100.0, 0.2, 2000.0)]
    fr = _ft4_frame(oracle, specs)
    cands, arr = oracle.ft4_candidates(fr, 200.0, 4000.0, 1.2, 200, want_arrays=True)
    heights = [c[2] for c in cands]
    assert heights == sorted(heights, reverse=True) and len(cands) >= 3
    # candidate frequency = interpolated peak of the 15-bin-smoothed spectrum - 1.5 tone spacings: the occupied band is
    # tone0 .. tone0 + 3*20.83 Hz, whose centre - 31.25 Hz is tone0; allow one smoothing half-width
    for audio_hz, _, _ in specs:
        assert any(abs(c[3] - audio_hz) <= 40.0 for c in cands[:6]), (audio_hz, cands[:6])
    assert (arr["sbase"][39:768] > 0).all()
    for b, _, h, fhz, _ in cands:
        assert 200.0 <= fhz <= 4910.0 and h >= np.float32(1.2) and 38 < b < 943
