"""Test helper: synthesize FT8-like 8-FSK bursts (Costas arrays at symbols 0/36/72, random data tones) as
complex IQ at a tuning offset, so that sync tests have real candidates to find.  Not a full FT8 encoder."""
import numpy as np

ICOS7 = [3, 1, 4, 0, 6, 5, 2]


def ft8_tones(rng):
    t = list(rng.integers(0, 8, 79))
    for base in (0, 36, 72):
        t[base:base + 7] = ICOS7
    return np.array(t)


def ft8_iq(fs, n, rf_hz, audio_hz, t0_s, amp, rng):
    """Complex baseband (relative to the LO) of one FT8 transmission: audio tone0 at `audio_hz` above the
    channel's USB dial offset rf_hz, starting t0_s into the buffer; 6.25 baud, 6.25 Hz tone spacing."""
    tones = ft8_tones(rng)
    sps = int(round(fs * 0.16))
    f = rf_hz + audio_hz + 6.25 * np.repeat(tones, sps)
    ph = 2 * np.pi * np.cumsum(f) / fs
    sig = amp * np.exp(1j * ph)
    out = np.zeros(n, np.complex64)
    i0 = int(round(t0_s * fs))
    m = min(len(sig), n - i0)
    out[i0:i0 + m] = sig[:m]
    return out


def ft4_iq(fs, n, rf_hz, audio_hz, t0_s, amp, rng):
    """FT4-like 4-FSK burst: 103 symbols of 48 ms, tone spacing 20.8333 Hz, tone 0 at audio_hz (no pulse shaping)."""
    tones = rng.integers(0, 4, 103)
    sps = int(round(fs * 0.048))
    f = rf_hz + audio_hz + (12000.0 / 576.0) * np.repeat(tones, sps)
    ph = 2 * np.pi * np.cumsum(f) / fs
    sig = amp * np.exp(1j * ph)
    out = np.zeros(n, np.complex64)
    i0 = int(round(t0_s * fs))
    m = min(len(sig), n - i0)
    out[i0:i0 + m] = sig[:m]
    return out
