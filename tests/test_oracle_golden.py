"""CPU: the oracle restatement (oracle/cwsl_oracle.c) against the fixtures generated from the compiled
reference headers (tests/gen_golden.py).  Everything here is bit-exact."""
import glob
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _u32(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "const_*.npz"))), ids=os.path.basename)
def test_constants(oracle, path):
    g = np.load(path)
    d = oracle.Demod(int(g["fs"]), float(np.float32(int(g["f"]))))
    assert d.block == int(g["block"]) and d.ntaps == int(g["ntaps"])
    assert np.array_equal(_u32(d.taps), g["taps_bits"])
    assert np.array_equal(_u32(d.tone), g["tone_bits"])
    assert np.array_equal(_u32(np.array([d.phase_inc])), g["inc_bits"])


def test_tap_landmarks(oracle):
    """SURVEY.md 8a-note sanity values for Fs=192 kHz."""
    t = oracle.Demod(192000, 0.0).taps
    assert t[0] == 0 and len(t) == 512
    assert abs(t[1] - (-9.81043013e-06)) < 1e-13
    assert abs(t[255] - 0.0312561318) < 1e-9 and t[255] == t[257]
    assert abs(t[256] - 0.0313074812) < 1e-9


_SLOTS = sorted(glob.glob(os.path.join(GOLD, "slot_*.npz")))


@pytest.mark.parametrize("path", _SLOTS, ids=os.path.basename)
def test_slot(oracle, path):
    g = np.load(path)
    mode, fs, f = str(g["mode"]), int(g["fs"]), int(g["f"])
    n_iq, iq_len = int(g["n_iq"]), int(g["iq_len"])
    iq = oracle.synth_iq(int(g["seed"]), n_iq, fs, tones_hz=list(g["tones"]), amp=float(g["amp"]))
    # demodulator level: audio + phasor trace
    d = oracle.Demod(fs, float(np.float32(f)))
    audio, trace = d.run(iq, trace=True)
    assert np.array_equal(trace[g["phasor_idx"]].view(np.uint64), g["phasor_bits"])
    assert np.array_equal(_u32(audio[:4096]), g["audio_head_bits"])
    assert np.array_equal(_u32(audio[-512:]), g["audio_tail_bits"])
    assert np.array_equal(_u32(audio[::997]), g["audio_every_bits"])
    assert _u32(np.array([np.abs(audio).max()], np.float32))[0] == g["audio_maxabs_bits"][0]
    assert oracle.checksum(audio) == float(g["audio_checksum"])
    # Instance level: framing + prepareAudio + int16 through the channel state machine
    c = oracle.Channel(mode, fs, iq_len, f)
    assert c.boundary(100) is None                     # first frame discarded (startEpochTime == 0)
    assert c.push_many(iq) == n_iq // iq_len
    r = c.boundary(100 + 15, want_f32=True)
    assert r["t_start"] == 100
    assert len(r["i16"]) == int(g["i16_len"])
    nv = int(g["n_valid"])
    assert np.array_equal(_u32(r["f32"][:nv]), _u32(audio)) and not r["f32"][nv:].any()
    assert _u32(np.array([r["factor"]]))[0] == g["factor_bits"][0]
    assert oracle.crc32(r["i16"]) == int(g["i16_crc32"])
    assert np.array_equal(r["i16"][:256], g["i16_head"])
    assert np.array_equal(r["i16"][nv - 256:nv], g["i16_tail"])
    assert not r["i16"][nv:].any()


def test_fixture_inventory():
    assert len(glob.glob(os.path.join(GOLD, "const_*.npz"))) == 15
    assert len(_SLOTS) == 12
