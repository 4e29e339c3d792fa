"""GPU: FT4 coherent sync (ft4sync_kernels.hpp) against oracle/ft4sync_oracle.c -- PARITY UNPINNED by the reference,
bit-identical to the repository's own restatement: frame spectrum, candidate baseband, refined records."""
import numpy as np
import pytest

from ft8_signal import ft4_iq, ICOS4

pytestmark = pytest.mark.gpu
FS, BLK = 48000, 1024


def _ft4_costas_iq(fs, n, rf_hz, audio_hz, t0_s, amp, rng):
    """FT4 burst with the real frame structure (Costas blocks at symbols 0/33/66/99) as complex IQ."""
    tones = np.array(rng.integers(0, 4, 103))
    for b, base in enumerate((0, 33, 66, 99)):
        tones[base:base + 4] = ICOS4[b]
    sps = int(round(fs * 0.048))
    f = rf_hz + audio_hz + (12000.0 / 576.0) * np.repeat(tones, sps)
    ph = 2 * np.pi * np.cumsum(f) / fs
    out = np.zeros(n, np.complex64)
    i0 = int(round(t0_s * fs))
    m = min(len(ph), n - i0)
    out[i0:i0 + m] = amp * np.exp(1j * ph[:m])
    return out


def _run_slot(ctx, oracle, bursts, seed, demod_hz=7000, coherent=True, max_cand=100):
    ctx.enable_sync(True, 1.5, max_cand, 200, 3000)
    ctx.enable_ft4_coherent(coherent)
    rx = ctx.receiver_open(FS, BLK, 0)
    ch = ctx.channel_open(rx, demod_hz, "FT4")
    n = int(7.5 * FS) // BLK * BLK
    rng = np.random.default_rng(seed)
    iq = oracle.synth_iq(seed, n, FS, tones_hz=[], amp=0.0) * 0.02
    for audio_hz, t0, amp in bursts:
        iq = iq + _ft4_costas_iq(FS, n, demod_hz, audio_hz, t0, amp, rng)
    ctx.slot_boundary("FT4", 10); ctx.push_iq(rx, iq.astype(np.complex64)); ctx.slot_boundary("FT4", 17)
    return ch, ctx.fetch_frame(ch)["i16"]


def test_frame_spectrum_bit_identical(ctx, oracle):
    ch, fr = _run_slot(ctx, oracle, [(1000.0, 0.7, 3000.0)], 11)
    got = ctx.sync_debug(ch, "ft4_cx")
    want = oracle.ft4_bigspec(fr)
    assert got.shape == want.shape == (36289,)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_candidate_baseband_and_records_bit_identical(ctx, oracle):
    bursts = [(1000.0, 0.70, 3000.0), (1900.0, 0.45, 2500.0), (2600.0, 1.10, 2800.0)]
    ch, fr = _run_slot(ctx, oracle, bursts, 12)
    cands = ctx.fetch_candidates(ch)
    want_c = oracle.ft4_candidates(fr, 200.0, 3000.0, 1.2, 100)
    assert cands and [tuple(c) for c in cands] == [tuple(c) for c in want_c]
    cx = oracle.ft4_bigspec(fr)
    cd0, _ = oracle.ft4_downsample(cx, np.float32(cands[0][3]))
    got_cd0 = ctx.sync_debug(ch, "ft4_cd0")
    assert np.array_equal(got_cd0.view(np.uint32), cd0.view(np.uint32))
    got = ctx.fetch_ft4_sync(ch)
    want = oracle.ft4_sync_all(fr, cands)
    assert len(got) == len(want) and len(want) >= 3
    for g, w in zip(got, want):
        assert g == w, (g, w)
    # ABI 5: the same slot in one call -- frame, list and refinements of one epoch under one ticket
    sl = ctx.fetch_slot(ch, max_list=100)
    assert sl["t_start"] == 10 and sl["list_kind"] == "FT4" and np.array_equal(sl["i16"], fr)
    assert [tuple(c) for c in sl["list"]] == [tuple(c) for c in cands] and sl["ft4_sync"] == got
    # and the refinement means something: each burst is found at its start time and frequency
    for audio_hz, t0, _ in bursts:
        near = [h for h in got if abs(h["f1_hz"] - audio_hz) <= 2.0 and abs(h["ibest"] / 666.67 - t0) <= 0.006]
        assert near and max(h["sync"] for h in near) > 2.5, (audio_hz, t0)


def test_early_and_late_bursts_exercise_the_edge_blocks(ctx, oracle):
    """Start times near the frame edges: segment 3 (negative start samples, truncated first block) and segment 2
    (truncated last block)."""
    bursts = [(800.0, 0.02, 3000.0), (2200.0, 1.45, 3000.0)]
    ch, fr = _run_slot(ctx, oracle, bursts, 13)
    cands = ctx.fetch_candidates(ch)
    got = ctx.fetch_ft4_sync(ch)
    want = oracle.ft4_sync_all(fr, cands)
    assert got == want and got
    assert {h["seg"] for h in got} >= {2, 3} or len({h["seg"] for h in got}) >= 2


def test_noise_only_and_disabled(ctx, oracle):
    ch, fr = _run_slot(ctx, oracle, [], 14)
    cands = ctx.fetch_candidates(ch)
    assert ctx.fetch_ft4_sync(ch) == oracle.ft4_sync_all(fr, cands)
    # coherent stage off: candidates still come, no refined records are produced
    ctx2 = type(ctx)(0)
    try:
        ch2, fr2 = _run_slot(ctx2, oracle, [(1000.0, 0.7, 3000.0)], 15, coherent=False)
        assert ctx2.fetch_candidates(ch2)
        assert ctx2.fetch_ft4_sync(ch2) is None
    finally:
        ctx2.close()
