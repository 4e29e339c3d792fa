"""GPU, exact mode (cwslg_set_exact): the reference's operation order on the GPU.  Everything is BIT-EXACT:
float audio vs the oracle (itself bit-identical to the compiled reference headers), int16 frames, and the
committed golden fixtures (audio samples, factor, int16 CRC32)."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GROUP = {"FT8": "FT8", "FT4": "FT4", "WSPR": "S120", "FST4W-120": "S120"}
FS, BLK = 192000, 2048


def test_exact_multi_channel_bitwise(ctx, oracle):
    ctx.set_exact(True)
    freqs = [0, 1234, 24000, 87000, -50000, -93000, -26000]
    na, nb = 24 * BLK, 120 * BLK
    tones = [f + d for f in freqs for d in (700.0, 2100.5)]
    iq = oracle.synth_iq(0xE5AC7, na + nb, FS, tones_hz=tones, amp=2.0e4)
    rx = ctx.receiver_open(FS, BLK, 0)
    chans = [ctx.channel_open(rx, f, "FT8") for f in freqs]
    ctx.push_iq(rx, iq[:na]); ctx.slot_boundary("FT8", 15)
    for k in range(na, na + nb, 7 * BLK):
        ctx.push_iq(rx, iq[k:min(k + 7 * BLK, na + nb)])
    ctx.slot_boundary("FT8", 30)
    ctx.push_iq(rx, iq[:nb]); ctx.slot_boundary("FT8", 45)          # a frame from a fresh demodulator too
    for f, ch in zip(freqs, chans):
        oc = oracle.Channel("FT8", FS, BLK, f)
        oc.push_many(iq[:na]); oc.boundary(15)
        oc.push_many(iq[na:]); oc.boundary(30)
        oc.push_many(iq[:nb]); r = oc.boundary(45, want_f32=True)
        a, nv = ctx.fetch_audio_f32(ch); g = ctx.fetch_frame(ch)
        assert np.array_equal(a.view(np.uint32), r["f32"].view(np.uint32)), f
        assert np.array_equal(g["i16"], r["i16"]), f
        assert np.float32(g["factor"]).view(np.uint32) == np.float32(r["factor"]).view(np.uint32)


@pytest.mark.parametrize("fs,block,mode", [(96000, 1024, "FT4"), (48000, 512, "FT8")])
def test_exact_other_rates(ctx, oracle, fs, block, mode):
    ctx.set_exact(True)
    f = 5000
    n1, n2 = 10 * block, 180 * block
    iq = oracle.synth_iq(17, n1 + n2, fs, tones_hz=[f + 900.0, f + 2100.0], amp=1.5e4)
    rx = ctx.receiver_open(fs, block, 0)
    ch = ctx.channel_open(rx, f, mode)
    oc = oracle.Channel(mode, fs, block, f)
    ctx.push_iq(rx, iq[:n1]); oc.push_many(iq[:n1])
    ctx.slot_boundary(GROUP[mode], 7); oc.boundary(7)
    ctx.push_iq(rx, iq[n1:]); oc.push_many(iq[n1:])
    ctx.slot_boundary(GROUP[mode], 14); r = oc.boundary(14, want_f32=True)
    a, _ = ctx.fetch_audio_f32(ch); g = ctx.fetch_frame(ch)
    assert np.array_equal(a.view(np.uint32), r["f32"].view(np.uint32))
    assert np.array_equal(g["i16"], r["i16"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "slot_*.npz"))), ids=os.path.basename)
def test_exact_golden_bitwise(ctx, oracle, path):
    """Against fixtures made from the compiled reference itself: audio bits, factor bits, int16 CRC32."""
    g = np.load(path)
    mode, fs, f = str(g["mode"]), int(g["fs"]), int(g["f"])
    n_iq, iq_len = int(g["n_iq"]), int(g["iq_len"])
    iq = oracle.synth_iq(int(g["seed"]), n_iq, fs, tones_hz=list(g["tones"]), amp=float(g["amp"]))
    ctx.set_exact(True)
    rx = ctx.receiver_open(fs, iq_len, 0)
    ch = ctx.channel_open(rx, f, mode)
    ctx.slot_boundary(GROUP[mode], 100)
    for k in range(0, n_iq, 128 * iq_len):
        ctx.push_iq(rx, iq[k:k + 128 * iq_len])
    ctx.slot_boundary(GROUP[mode], 115)
    a, nv = ctx.fetch_audio_f32(ch); fr = ctx.fetch_frame(ch)
    audio = a[:nv]
    assert np.array_equal(audio[:4096].view(np.uint32), g["audio_head_bits"])
    assert np.array_equal(audio[-512:].view(np.uint32), g["audio_tail_bits"])
    assert np.array_equal(audio[::997].view(np.uint32), g["audio_every_bits"])
    assert oracle.checksum(audio) == float(g["audio_checksum"])
    assert np.float32(fr["factor"]).view(np.uint32) == g["factor_bits"][0]
    assert oracle.crc32(fr["i16"]) == int(g["i16_crc32"])


def test_exact_and_fast_agree_within_tolerance(ctx, oracle):
    f, n = -26000, 200 * BLK
    iq = oracle.synth_iq(3, n, FS, tones_hz=[f + 1000.0, f + 1800.0], amp=2e4)
    out = []
    for exact in (False, True):
        ctx.set_exact(exact)
        rx = ctx.receiver_open(FS, BLK, 0)
        ch = ctx.channel_open(rx, f, "FT8")
        ctx.slot_boundary("FT8", 1); ctx.push_iq(rx, iq); ctx.slot_boundary("FT8", 2)
        out.append(ctx.fetch_audio_f32(ch)[0])
        ctx.receiver_close(rx)
    peak = np.abs(out[1]).max()
    assert np.abs(out[0].astype(np.float64) - out[1]).max() <= 1e-5 * peak
