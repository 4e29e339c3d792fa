"""Adversarial inputs of the demodulator: frames whose own peak is small next to the input level, carriers at
the band edge, full-scale and all-zero IQ, a peak inside the filter's start-up transient.  Shared by tests/gen_golden.py (which makes
tests/golden/adv_*.npz from the compiled reference headers, oracle/_ref) and the tests that read those fixtures.

The inputs are built with numpy in double and rounded once to float32; a fixture stores the CRC32 of its input so that a test can tell
"different libm, different input" from "different output".  Channel: 192 kHz, FT8, demod_hz = -26000 (BASELINE configs[0]); the
passband is [F, F + 6000] Hz = [-26000, -20000]."""
import zlib

import numpy as np

FS, BLK, F = 192000, 2048, -26000
N = 200 * BLK                      # 409 600 complex samples -> 25 600 audio samples
NOISE = 0.5                        # per component: an "empty" band is noise 92 dB below the carriers


def _carrier(t, hz, amp):
    return amp * np.exp(2j * np.pi * hz * t)


CASES = {
    # name: (carriers [(Hz, amplitude)], noise sigma, post-processing)
    "oob1":       ([(41000.0, 2.0e4)], NOISE, None),
    "oob2":       ([(41000.0, 2.0e4), (-70000.0, 2.0e4)], NOISE, None),
    "oob8":       ([(41000.0, 1.0e4), (-70000.0, 1.0e4), (12000.0, 1.0e4), (-55555.0, 1.0e4), (88000.0, 1.0e4), (-93000.0, 1.0e4),
                    (3000.0, 1.0e4), (-40000.0, 1.0e4)], NOISE, None),
    "edge_below": ([(F - 50.0, 2.0e4)], NOISE, None),            # 50 Hz outside the lower band edge
    "edge_in_lo": ([(F + 50.0, 2.0e4)], NOISE, None),            # 50 Hz inside it
    "edge_in_hi": ([(F + 6000.0 - 50.0, 2.0e4)], NOISE, None),
    "edge_above": ([(F + 6000.0 + 50.0, 2.0e4)], NOISE, None),
    "fullscale":  ([(F + 1500.0, 3.0e4), (41000.0, 3.0e4)], 100.0, "int"),      # CWSL hands over integer-derived samples, |x| ~ 3e4
    "zeros":      ([], 0.0, "zero"),                              # factor = 32767 / (0 + 1)
    "early_peak": ([(F + 1500.0, 2.0e4)], NOISE, "burst"),        # in-band carrier during the first 320 samples only: the frame's
                                                                  # peak sits in the first 31 outputs (window not yet full)
}


def make_iq(name):
    carriers, sigma, post = CASES[name]
    t = np.arange(N) / FS
    rng = np.random.default_rng(20260101 + sorted(CASES).index(name))
    x = np.zeros(N, np.complex128)
    for k, (hz, amp) in enumerate(carriers):
        c = _carrier(t, hz, amp)
        if post == "burst":
            c[320:] = 0.0
        x += c
    if sigma:
        x += (rng.standard_normal(N) + 1j * rng.standard_normal(N)) * sigma
    if post == "int":
        x = np.round(x.real) + 1j * np.round(x.imag)
    if post == "zero":
        x[:] = 0.0
    return x.astype(np.complex64)


def iq_crc(iq):
    return zlib.crc32(np.ascontiguousarray(iq).view(np.uint8).tobytes()) & 0xFFFFFFFF
