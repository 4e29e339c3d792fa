"""CPU: slot framing / normalise / int16 / WAV rules of Instance.cpp, DecoderPool.hpp, WaveFile.hpp as the
oracle restates them (these rows have no compilable reference; the asserts below are the quoted lines)."""
import struct

import numpy as np
import pytest

FS, BLK = 192000, 2048


def test_frame_lengths(oracle):
    # Instance.cpp:149 : 12000 * (period + 5)
    want = {"FT8": 240000, "JS8": 240000, "FT4": 150000, "WSPR": 1500000, "FST4W-120": 1500000, "Q65-30": 420000,
            "JT65": 780000, "FST4-60": 780000, "FST4-300": 3660000, "FST4W-900": 10860000, "FST4-1800": 21660000}
    for m, n in want.items():
        assert oracle.frame_len(m) == n
    assert oracle.frame_len("PSK31") == 0


def test_prepare_audio_rules(oracle):
    # maxVal over the whole frame, |minVal| wins if larger, factor = 32767/(max+1) * scale (Instance.cpp:294-329)
    buf = np.array([0, 10, -40, 5, 0, 0], np.float32)
    s, f, pk = oracle.prepare_audio(buf, "FT8")
    assert pk == 40 and f == np.float32(np.float32(32767.0) / np.float32(41.0)) * np.float32(0.90)
    assert np.array_equal(s, buf * f)
    _, fw, _ = oracle.prepare_audio(buf, "WSPR")
    assert fw == np.float32(np.float32(32767.0) / np.float32(41.0)) * np.float32(0.20)
    _, f4, _ = oracle.prepare_audio(buf, "FST4W-120")           # not "WSPR": FT factor (:320 exact compare)
    assert f4 == f
    # all-negative and all-positive frames: the rule still yields max|x|
    assert oracle.prepare_audio(np.array([-3, -7, -5], np.float32), "FT8")[2] == 7
    assert oracle.prepare_audio(np.array([3, 7, 5], np.float32), "FT8")[2] == 7


def test_int16_rounding_rule(oracle):
    # (int16)(x + 0.5f): add then truncate toward zero (Instance.cpp:240)
    x = np.array([0.0, 0.4, 0.5, 0.6, 1.5, -0.4, -0.5, -0.7, -1.4, -1.5, -1.6, -2.5, 29490.3, -29490.7], np.float32)
    want = np.array([0, 0, 1, 1, 2, 0, 0, 0, 0, -1, -1, -2, 29490, -29490], np.int16)
    assert np.array_equal(oracle.to_int16(x), want)


def test_first_frame_discarded_and_no_demod_reset(oracle):
    """Instance.cpp:224-227: startEpochTime==0 -> `continue` BEFORE the SSBD re-creation at :251, so the
    first emitted frame continues filter history and phasor across the first boundary."""
    f = 1234
    na, nb = 6 * BLK, 20 * BLK
    iq = oracle.synth_iq(1, na + nb, FS, tones_hz=[f + 1000.0], amp=1e4)
    c = oracle.Channel("FT8", FS, BLK, f)
    c.push_many(iq[:na])
    assert c.boundary(15) is None
    c.push_many(iq[na:])
    r = c.boundary(30, want_f32=True)
    cont = oracle.Demod(FS, f).run(iq)                    # one demodulator over both pieces
    assert np.array_equal(r["f32"][:nb // 16].view(np.uint32), cont[na // 16:].view(np.uint32))
    fresh = oracle.Demod(FS, f).run(iq[na:])
    assert not np.array_equal(r["f32"][:nb // 16], fresh)
    # the NEXT frame does start from a fresh demodulator (:251)
    c.push_many(iq[:nb])
    r2 = c.boundary(45, want_f32=True)
    assert np.array_equal(r2["f32"][:nb // 16].view(np.uint32), oracle.Demod(FS, f).run(iq[:nb]).view(np.uint32))
    assert r["t_start"] == 15 and r2["t_start"] == 30


def test_overflow_guard(oracle):
    # Instance.cpp:268 : fill + iq_len > size-1 -> block dropped (IQ length compared with audio samples)
    c = oracle.Channel("FT4", FS, BLK, 0)
    blk = np.zeros(BLK, np.complex64)
    took = 0
    for _ in range(1200):
        took += c.push(blk)
    # accepted while fill + 2048 <= 149999, fill advances by 128
    assert took == (150000 - 1 - BLK) // 128 + 1 == 1156
    assert c.dropped == 1200 - took and c.fill == took * 128


def test_trailing_partial_block(oracle):
    f = -26000
    n = 5 * BLK + 512
    iq = oracle.synth_iq(4, n, FS, tones_hz=[f + 900.0], amp=1e4)
    c = oracle.Channel("FT8", FS, BLK, f)
    c.boundary(1)
    assert c.push_stream(iq) == 6
    r = c.boundary(2, want_f32=True)
    assert np.array_equal(r["f32"][:n // 16].view(np.uint32), oracle.Demod(FS, f).run(iq).view(np.uint32))


def test_wav_header(oracle, tmp_path):
    # WaveFile.hpp:19-35,96-113 : 46-byte header, 18-byte WAVEFORMATEX, PCM mono 12 kHz int16
    h = oracle.wav_header(240000)
    assert len(h) == 46
    riff, flen, wave, fmt, fmtlen, tag, ch, sr, bps, align, bits, cb, data, dlen = struct.unpack("<4sI4s4sIHHIIHHH4sI", h)
    assert (riff, wave, fmt, data) == (b"RIFF", b"WAVE", b"fmt ", b"data")
    assert (fmtlen, tag, ch, sr, bps, align, bits, cb) == (18, 1, 1, 12000, 24000, 2, 16, 0)
    assert dlen == 480000 and flen == 46 + 480000 - 8
    p = str(tmp_path / "x.wav")
    pcm = (np.arange(1000) - 500).astype(np.int16)
    assert oracle.lib().orc_wav_write(p.encode(), pcm, 1000) == 0
    raw = open(p, "rb").read()
    assert raw[:46] == oracle.wav_header(1000) and np.array_equal(np.frombuffer(raw[46:], np.int16), pcm)


def test_synth_generator_is_exact_integers(oracle):
    x = oracle.synth_iq(123, 4096).view(np.float32)
    assert np.array_equal(x * 32, np.round(x * 32))            # k/32 values: exact in any IEEE float
    assert 1000 < x.std() < 1400
    y = oracle.synth_iq(123, 2048, first=2048).view(np.float32)
    assert np.array_equal(y, x[4096:])                         # counter-based: any window regenerates
