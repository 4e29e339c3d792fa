"""GPU parity against the committed golden fixtures (tests/golden/, generated from the compiled
reference headers by tests/gen_golden.py): phasor checkpoints bit-exact, float audio within 1e-5 of
frame peak at every stored sample, peak/factor within float rounding, int16 within the documented tie rule."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SLOTS = sorted(glob.glob(os.path.join(GOLD, "slot_*.npz")))
GROUP = {"FT8": "FT8", "FT4": "FT4", "WSPR": "S120", "FST4W-120": "S120"}


@pytest.mark.parametrize("path", SLOTS, ids=os.path.basename)
def test_slot_fixture(ctx, oracle, path):
    g = np.load(path)
    mode, fs, f = str(g["mode"]), int(g["fs"]), int(g["f"])
    n_iq, iq_len = int(g["n_iq"]), int(g["iq_len"])
    iq = oracle.synth_iq(int(g["seed"]), n_iq, fs, tones_hz=list(g["tones"]), amp=float(g["amp"]))  # input generator only
    rx = ctx.receiver_open(fs, iq_len, 0)
    ch = ctx.channel_open(rx, f, mode)
    ctx.slot_boundary(GROUP[mode], 100)                  # discarded first frame; demodulator keeps running
    step = 128 * iq_len
    for k in range(0, n_iq, step):
        ctx.push_iq(rx, iq[k:k + step])
    ctx.slot_boundary(GROUP[mode], 115)
    a, nv = ctx.fetch_audio_f32(ch)
    fr = ctx.fetch_frame(ch)
    assert nv == int(g["n_valid"]) and len(fr["i16"]) == int(g["i16_len"]) and fr["t_start"] == 100
    audio = a[:nv]
    peak = float(g["audio_maxabs_bits"].view(np.float32)[0])
    tol = 1e-5 * peak
    head = g["audio_head_bits"].view(np.float32); tail = g["audio_tail_bits"].view(np.float32)
    every = g["audio_every_bits"].view(np.float32)
    errs = [np.abs(audio[:4096].astype(np.float64) - head).max(), np.abs(audio[-512:].astype(np.float64) - tail).max(),
            np.abs(audio[::997].astype(np.float64) - every).max()]
    assert max(errs) <= tol, (errs, tol)
    assert abs(float(np.abs(audio).max()) - peak) <= tol
    assert abs(oracle.checksum(audio) - float(g["audio_checksum"])) <= tol * 126 * nv      # worst-case bound of the weighted sum
    fac = float(g["factor_bits"].view(np.float32)[0])
    assert abs(float(fr["factor"]) - fac) <= 2e-6 * fac
    for got, want in ((fr["i16"][:256], g["i16_head"]), (fr["i16"][nv - 256:nv], g["i16_tail"])):
        assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1
    assert not fr["i16"][nv:].any()
    # phasor: every stored index that is a checkpoint (multiple of the stride) must be bit-exact
    idx = g["phasor_idx"]; bits = g["phasor_bits"]
    st = ctx.checkpoint_stride()
    ck = ctx.phasor_checkpoints(ch, int(idx.max()) // st + 1)
    sel = idx % st == 0
    assert sel.sum() >= 3
    assert np.array_equal(ck[idx[sel] // st].view(np.uint64), bits[sel])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "const_*.npz"))), ids=os.path.basename)
def test_constants_fixture(ctx, path):
    g = np.load(path)
    rx = ctx.receiver_open(int(g["fs"]), 64 * (int(g["fs"]) // 12000), 0)
    ch = ctx.channel_open(rx, int(g["f"]), "FT8")
    taps, tone, inc = ctx.channel_constants(ch)
    assert np.array_equal(taps.view(np.uint32), g["taps_bits"])
    assert np.array_equal(tone.view(np.uint32), g["tone_bits"])
    assert np.array_equal(np.array([inc]).view(np.uint32), g["inc_bits"])
