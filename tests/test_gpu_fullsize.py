"""GPU: the BASELINE.json configurations at FULL size (the driver's `-m gpu` run sees them, not only the builder).

configs[1]  64 FT8 slots x 15 s on one GPU
configs[2]  768 FT8 + 256 FT4 slots, two slot clocks (FT4 fires twice per FT8 slot), sync stage on
configs[4]  128 WSPR + 128 FST4W-120 slots x 120 s (47 GB of IQ resident)
north_star  4096 FT8 slots x 15 s resident on ONE MI355X (94 GB of IQ)
Every slot's IQ comes from the device-side synthetic source, which is bit-identical to the oracle's generator, so a few
slots spread over the range are re-derived on the CPU and checked in full: float audio within 1e-5 of frame peak,
int16 equal up to +-1 LSB rounding ties, FT8/FT4/WSPR/FST4W candidate lists bit-identical to the restatement run on the GPU frame.
Size-independent properties cover ALL slots: frame counts, valid-sample counts, a finite non-zero scale factor.
"""
import numpy as np
import pytest

from conftest import assert_frames_match, assert_int16_match

pytestmark = pytest.mark.gpu
FS, BLK = 192000, 2048
PERIOD = {"FT8": 15, "FT4": 7.5, "WSPR": 120, "FST4W-120": 120}
GROUP = {"FT8": "FT8", "FT4": "FT4", "WSPR": "S120", "FST4W-120": "S120"}


def _freq(gs):
    return -90000 + (gs * 4373) % 176000


def _tones(f, gs):
    # one carrier inside the 120 s modes' search windows (1500 +- 110 Hz, 1400..1600 Hz), two elsewhere in the passband
    return [f + 600.0 + 37.0 * (gs % 11), f + 1500.0 - 60.0 + 9.0 * (gs % 13), f + 2450.0 - 13.0 * (gs % 7)]


def _fetch_cands(ctx, ch, mode):
    if mode in ("FT8", "FT4"):
        return ctx.fetch_candidates(ch, 200)
    return ctx.fetch_wspr_candidates(ch) if mode == "WSPR" else ctx.fetch_fst4w_candidates(ch)


def _oracle_cands(oracle, frame, mode):
    if mode == "FT8":
        return oracle.ft8_sync(frame, 200, 3000, 1.5, 200)
    if mode == "FT4":
        return oracle.ft4_candidates(frame, 200.0, 3000.0, 1.2, 200)
    return oracle.wspr_search(frame) if mode == "WSPR" else oracle.fst4w_candidates(frame)


def _bits(c):
    return tuple(np.float32(x).view(np.uint32) if isinstance(x, float) else x for x in c)


def _run(ctx, oracle, modes, check, sync=False):
    """modes: list of mode names, one per slot.  One step = the longest period present; shorter modes fire their own
    boundaries inside it (FT4: two frames per FT8 slot)."""
    longest = max(PERIOD[m] for m in modes)
    n_long = int(longest * FS) // BLK * BLK
    cap = n_long + 4 * BLK
    if sync:
        ctx.enable_sync(True, 1.5, 200, 200, 3000)
        ctx.enable_long_sync(True)
    slots = []
    for gs, mode in enumerate(modes):
        f = _freq(gs)
        rx = ctx.receiver_open(FS, BLK, 0, ring_blocks=cap // BLK)
        # the whole ring is filled before any channel exists (ring index = sample index of the generator); the timed part
        # only commits what is already resident, so the stream the channel sees is ring[0:n_long]
        ctx.push_synth(rx, 0xC0FFEE ^ gs, cap // 2, BLK, tones_hz=_tones(f, gs), amp=2.0e4)
        ctx.push_synth(rx, 0xC0FFEE ^ gs, cap - cap // 2, BLK, tones_hz=_tones(f, gs), amp=2.0e4)
        ch = ctx.channel_open(rx, f, mode)
        slots.append((rx, ch, mode, f, gs))
    for g in sorted({GROUP[m] for m in modes}):
        ctx.slot_boundary(g, 100)                              # every clock fires at t = 0: partial-slot frames, discarded
    # one long period of IQ, committed in the pieces the shortest clock needs
    shortest = min(PERIOD[m] for m in modes)
    parts = int(round(longest / shortest))
    piece = n_long // parts // BLK * BLK
    frames = {}
    for part in range(parts):
        ctx.ring_commit_all(piece, BLK)
        for g in sorted({GROUP[m] for m in modes}):
            per = min(PERIOD[m] for m in modes if GROUP[m] == g)
            if (part + 1) % int(round(per / shortest)) == 0:
                ctx.slot_boundary(g, 100 + int((part + 1) * shortest))
        if part == 0:
            for k in check:
                rx, ch, mode, f, gs = slots[k]
                if PERIOD[mode] == shortest and parts > 1:
                    frames[(k, 0)] = (ctx.fetch_frame(ch), ctx.fetch_audio_f32(ch), _fetch_cands(ctx, ch, mode) if sync else None)
    ctx.synchronize()
    st = ctx.stats()
    n_emit = sum(int(round(longest / PERIOD[m])) for m in modes)
    assert st["frames_emitted"] == n_emit and st["frames_discarded"] == len(modes) and st["blocks_dropped"] == 0
    # every slot: a whole frame of the right length came out
    for rx, ch, mode, f, gs in slots:
        g = ctx.fetch_frame(ch)
        per_samples = piece * int(round(PERIOD[mode] / shortest))
        assert g is not None and g["n_valid"] == per_samples // 16 and len(g["i16"]) == int(12000 * (PERIOD[mode] + 5))
        assert np.isfinite(g["factor"]) and 0 < g["factor"] < 1e3
    # a few slots in full against the oracle
    worst = 0.0
    for k in check:
        rx, ch, mode, f, gs = slots[k]
        reps = int(round(longest / PERIOD[mode]))
        per_samples = piece * int(round(PERIOD[mode] / shortest))
        iq = oracle.synth_iq(0xC0FFEE ^ gs, n_long, FS, tones_hz=_tones(f, gs), amp=2.0e4)
        oc = oracle.Channel(mode, FS, BLK, f)
        assert oc.boundary(100) is None
        for rep in range(reps):
            oc.push_many(iq[rep * per_samples:(rep + 1) * per_samples])
            r = oc.boundary(100 + int((rep + 1) * PERIOD[mode]), want_f32=True)
            if rep == reps - 1:
                got = (ctx.fetch_frame(ch), ctx.fetch_audio_f32(ch), _fetch_cands(ctx, ch, mode) if sync else None)
            elif (k, rep) in frames:
                got = frames[(k, rep)]
            else:
                continue
            g, (a, nv), cands = got
            assert g["t_start"] == r["t_start"] and nv == per_samples // 16
            worst = max(worst, assert_frames_match(a, r["f32"]))
            assert_int16_match(g["i16"], r["i16"], r["f32"] * r["factor"])
            if cands is not None:
                ref = _oracle_cands(oracle, g["i16"], mode)
                assert [_bits(c) for c in cands] == [_bits(c) for c in ref]
    return worst


def test_config1_64_ft8_slots(ctx, oracle):
    _run(ctx, oracle, ["FT8"] * 64, check=[0, 21, 42, 63])


def test_config2_768_ft8_256_ft4_with_sync(ctx, oracle):
    modes = ["FT8"] * 768 + ["FT4"] * 256
    _run(ctx, oracle, modes, check=[0, 401, 767, 768, 900, 1023], sync=True)


def test_config4_128_wspr_128_fst4w(ctx, oracle):
    modes = ["WSPR" if s % 2 == 0 else "FST4W-120" for s in range(256)]
    _run(ctx, oracle, modes, check=[0, 85, 170, 255], sync=True)      # incl. the WSPR / FST4W-120 candidate lists


def test_north_star_4096_ft8_slots_on_one_gpu(ctx, oracle):
    _run(ctx, oracle, ["FT8"] * 4096, check=[0, 1365, 2730, 4095], sync=True)
