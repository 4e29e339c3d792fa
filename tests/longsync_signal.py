"""Test helpers: WSPR- and FST4W-like 4-FSK audio (real 12 kHz) so that the 120 s candidate searches have something to find.
Not encoders: the WSPR helper puts the real 162-symbol sync vector on random data bits, the FST4W helper sends random tones."""
import numpy as np

PR3 = [1,1,0,0,0,0,0,0,1,0,0,0,1,1,1,0,0,0,1,0, 0,1,0,1,1,1,1,0,0,0,0,0,0,0,1,0,0,1,0,1,
       0,0,0,0,0,0,1,0,1,1,0,0,1,1,0,1,0,0,0,1, 1,0,1,0,0,0,0,1,1,0,1,0,1,0,1,0,1,0,0,1,
       0,0,1,0,1,1,0,0,0,1,1,0,1,0,1,0,0,0,1,0, 0,0,0,0,1,0,0,1,0,0,1,1,1,0,1,1,0,0,1,1,
       0,1,0,0,0,1,1,1,0,0,0,0,0,1,0,1,0,0,1,1, 0,0,0,0,0,0,0,1,1,0,1,0,1,1,0,0,0,1,1,0,
       0,0]
assert len(PR3) == 162


def _fsk(n, f_of_t, t0, amp):
    ph = 2 * np.pi * np.cumsum(f_of_t) / 12000.0
    out = np.zeros(n)
    i0 = int(round(t0 * 12000))
    m = min(len(ph), n - i0)
    out[i0:i0 + m] = amp * np.cos(ph[:m])
    return out


def wspr_audio(n, f0_hz, t0_s, amp, rng, drift_hz=0.0):
    """tone 0 at f0_hz, spacing 12000/8192 Hz, 8192 samples per symbol, symbol = sync + 2 data"""
    sym = np.array(PR3) + 2 * rng.integers(0, 2, 162)
    f = f0_hz + (12000.0 / 8192.0) * np.repeat(sym, 8192)
    tt = np.arange(len(f)) / 12000.0
    f = f + drift_hz * (tt - tt.mean()) / (tt[-1] - tt[0])
    return _fsk(n, f, t0_s, amp)


def fst4w_audio(n, f0_hz, t0_s, amp, rng):
    """160 random 4-FSK symbols of 8200 samples, tone spacing = baud = 12000/8200 Hz"""
    sym = rng.integers(0, 4, 160)
    f = f0_hz + (12000.0 / 8200.0) * np.repeat(sym, 8200)
    return _fsk(n, f, t0_s, amp)


def to_i16(a):
    return np.clip(np.round(a), -32768, 32767).astype(np.int16)


def fsk_iq(fs, n, rf_hz, audio_f0_hz, t0_s, amp, sym, sps_audio, spacing_hz, drift_hz=0.0):
    """Complex IQ (relative to the LO) of a 4-FSK transmission whose tone 0 lands at audio_f0_hz in the USB channel tuned to
    rf_hz; sps_audio = samples per symbol at 12 kHz."""
    sps = sps_audio * (fs // 12000)
    nsig = len(sym) * sps
    out = np.zeros(n, np.complex64)
    i0 = int(round(t0_s * fs))
    m = min(nsig, n - i0)
    # phase accumulated in float64 in pieces (a 120 s transmission at 192 kHz is 21 M samples)
    ph0 = 0.0
    for a in range(0, m, 1 << 22):
        b = min(m, a + (1 << 22))
        k = np.arange(a, b)
        f = rf_hz + audio_f0_hz + spacing_hz * sym[k // sps] + drift_hz * (k / nsig - 0.5)
        ph = ph0 + 2 * np.pi * np.cumsum(f) / fs
        out[i0 + a:i0 + b] = (amp * np.exp(1j * ph)).astype(np.complex64)
        ph0 = float(ph[-1]) % (2 * np.pi)
    return out


def wspr_iq(fs, n, rf_hz, audio_f0_hz, t0_s, amp, rng, drift_hz=0.0):
    sym = np.array(PR3) + 2 * rng.integers(0, 2, 162)
    return fsk_iq(fs, n, rf_hz, audio_f0_hz, t0_s, amp, sym, 8192, 12000.0 / 8192.0, drift_hz)


def fst4w_iq(fs, n, rf_hz, audio_f0_hz, t0_s, amp, rng):
    return fsk_iq(fs, n, rf_hz, audio_f0_hz, t0_s, amp, rng.integers(0, 4, 160), 8200, 12000.0 / 8200.0)
