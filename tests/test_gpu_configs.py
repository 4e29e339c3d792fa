"""GPU: the BASELINE.json configurations at test scale.

configs[2]  mixed FT8/FT4 slots on one GPU: two slot-clock groups with different periods, boundaries interleaved
            exactly as the wall-clock threads would fire them (FT4 every 7.5 s, FT8 every 15 s).
configs[4]  WSPR + FST4W-120 long-integration path: 120 s frames (1.5 M samples), the 0.20 scale for "WSPR" only.
configs[3]  sharded slots: the per-rank shard of a 2-rank job reproduces the unsharded result slot for slot.
Inputs come from the device-side synthetic source (bit-identical to the oracle's generator), so the oracle can
re-derive every slot's IQ from its seed.
"""
import numpy as np
import pytest

from conftest import assert_frames_match, assert_int16_match

pytestmark = pytest.mark.gpu
FS, BLK = 192000, 2048


def _tones(f):
    return [f + 650.0, f + 1490.0, f + 2310.0]


def test_config2_mixed_ft8_ft4(ctx, oracle):
    n_ft8, n_ft4 = 12, 6
    half = 1440000 // BLK * BLK                    # 7.5 s worth of whole blocks
    slots = []
    for s in range(n_ft8 + n_ft4):
        mode = "FT8" if s < n_ft8 else "FT4"
        f = -88000 + 9700 * s
        rx = ctx.receiver_open(FS, BLK, 0, ring_blocks=2 * half // BLK + 8)
        ch = ctx.channel_open(rx, f, mode)
        slots.append((rx, ch, mode, f, 0x5EED ^ s))
    ctx.slot_boundary("FT8", 100); ctx.slot_boundary("FT4", 100)       # both clocks fire at t=0: discard
    # 7.5 s of IQ, FT4 boundary; 7.5 s more, FT4 + FT8 boundaries
    got4 = []
    for part in range(2):
        for rx, ch, mode, f, seed in slots:
            ctx.push_synth(rx, seed, half, BLK, tones_hz=_tones(f), amp=1.5e4)
        ctx.slot_boundary("FT4", 107 + 8 * part)
        got4.append({ch: (ctx.fetch_frame(ch), ctx.fetch_audio_f32(ch)) for _, ch, m, _, _ in slots if m == "FT4"})
    ctx.slot_boundary("FT8", 115)
    for rx, ch, mode, f, seed in slots:
        iq = oracle.synth_iq(seed, 2 * half, FS, tones_hz=_tones(f), amp=1.5e4)
        oc = oracle.Channel(mode, FS, BLK, f)
        assert oc.boundary(100) is None
        if mode == "FT4":
            for part in range(2):
                oc.push_many(iq[part * half:(part + 1) * half])
                r = oc.boundary(107 + 8 * part, want_f32=True)
                g, (a, nv) = got4[part][ch]
                assert g["t_start"] == r["t_start"] and nv == half // 16 and len(g["i16"]) == 150000
                assert_frames_match(a, r["f32"])
                assert_int16_match(g["i16"], r["i16"], r["f32"] * r["factor"])
        else:
            oc.push_many(iq)
            r = oc.boundary(115, want_f32=True)
            g = ctx.fetch_frame(ch); a, nv = ctx.fetch_audio_f32(ch)
            assert g["t_start"] == 100 and nv == 2 * half // 16 and len(g["i16"]) == 240000
            assert_frames_match(a, r["f32"])
            assert_int16_match(g["i16"], r["i16"], r["f32"] * r["factor"])
    st = ctx.stats()
    assert st["frames_emitted"] == n_ft8 + 2 * n_ft4 and st["frames_discarded"] == n_ft8 + n_ft4


def test_config4_wspr_fst4w_long_frames(ctx, oracle):
    n = 23040000 // BLK * BLK                      # 120 s
    slots = []
    for s in range(6):
        mode = "WSPR" if s % 2 == 0 else "FST4W-120"
        f = -60000 + 21000 * s
        rx = ctx.receiver_open(FS, BLK, 0, ring_blocks=n // BLK + 4)
        ch = ctx.channel_open(rx, f, mode)
        slots.append((rx, ch, mode, f, 0xABCD ^ s))
    ctx.slot_boundary("S120", 120)
    for rx, ch, mode, f, seed in slots:
        ctx.push_synth(rx, seed, n, BLK, tones_hz=[f + 1500.0, f + 1502.9], amp=8.0e3)
    ctx.slot_boundary("S120", 240)
    factors = {}
    for k, (rx, ch, mode, f, seed) in enumerate(slots):
        g = ctx.fetch_frame(ch)
        assert len(g["i16"]) == 1500000 and g["n_valid"] == n // 16 and g["t_start"] == 120
        factors[mode] = float(g["factor"])
        if k < 2:                                   # full-frame oracle check for one slot of each mode
            iq = oracle.synth_iq(seed, n, FS, tones_hz=[f + 1500.0, f + 1502.9], amp=8.0e3)
            oc = oracle.Channel(mode, FS, BLK, f)
            oc.boundary(120); oc.push_many(iq)
            r = oc.boundary(240, want_f32=True)
            a, nv = ctx.fetch_audio_f32(ch)
            assert_frames_match(a, r["f32"])
            assert_int16_match(g["i16"], r["i16"], r["f32"] * r["factor"])
            assert abs(float(g["factor"]) - float(r["factor"])) <= 2e-6 * float(r["factor"])
    assert factors["WSPR"] < 0.3 * factors["FST4W-120"]      # 0.20 vs 0.90 (Instance.cpp:320)


def test_config3_shard_equals_unsharded(ctx, oracle):
    """Rank r of a 2-rank job owns slots [r*S,(r+1)*S): running only that shard gives the same frames as
    running all slots together (no cross-slot arithmetic anywhere)."""
    from cwsl_digi_amd import shard
    total, n = 8, 64 * BLK

    def run(slot_ids):
        out = {}
        chans = []
        for gs in slot_ids:
            f = -90000 + (gs * 4373) % 176000
            rx = ctx.receiver_open(FS, BLK, 0, ring_blocks=n // BLK + 8)
            ch = ctx.channel_open(rx, f, "FT8")
            ctx.slot_boundary_channel(ch, 1)
            ctx.push_synth(rx, 0xC0FFEE ^ gs, n, BLK, tones_hz=_tones(f), amp=2e4)
            chans.append((gs, rx, ch))
        ctx.slot_boundary("FT8", 16)
        for gs, rx, ch in chans:
            out[gs] = ctx.fetch_frame(ch)["i16"].copy()
            ctx.receiver_close(rx)
        return out

    full = run(range(total))
    for rank in range(2):
        part = run(shard.slots_of_rank(total, rank, 2))
        assert sorted(part) == list(shard.slots_of_rank(total, rank, 2))
        for gs, fr in part.items():
            assert np.array_equal(fr, full[gs])
