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


ICOS4 = [[0, 1, 3, 2], [1, 0, 2, 3], [2, 3, 1, 0], [3, 2, 0, 1]]


def ft4_frame_tones(rng):
    """103 FT4 channel symbols: Costas blocks at symbols 0, 33, 66, 99, random data elsewhere."""
    t = list(rng.integers(0, 4, 103))
    for b, base in enumerate((0, 33, 66, 99)):
        t[base:base + 4] = ICOS4[b]
    return np.array(t)


def ft4_audio(n, f0_hz, t_start_s, amp, rng, fs=12000):
    """Real 12 kHz audio of one FT4 transmission whose first Costas symbol starts t_start_s into the buffer
    (tone 0 at f0_hz, 20.833 Hz spacing, 48 ms symbols, continuous phase, no pulse shaping)."""
    tones = ft4_frame_tones(rng)
    sps = 576 * fs // 12000
    f = f0_hz + (12000.0 / 576.0) * np.repeat(tones, sps)
    ph = 2 * np.pi * np.cumsum(f) / fs
    out = np.zeros(n)
    i0 = int(round(t_start_s * fs))
    m = min(len(ph), n - i0)
    out[i0:i0 + m] = amp * np.cos(ph[:m])
    return out
