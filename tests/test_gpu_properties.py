"""GPU: size-independent properties of the hot path at BASELINE's full slot size (the oracle is not needed, so these run
where an oracle comparison would take too long): exact power-of-two scaling, independence of push granularity, channel
independence on a shared receiver, and agreement of the fast and the exact mode within the stated tolerance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FS, BLK = 192000, 2048
SLOT = 2880000 // BLK * BLK            # whole Receiver blocks of one 15 s FT8 slot


def _slot_iq(seed, n=SLOT):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    iq = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 700.0
    for f, a in ((-26000 + 1200.0, 9000.0), (41000 + 800.0, 6000.0), (-77000 + 2100.0, 12000.0)):
        iq += a * np.exp(2j * np.pi * f * t / FS)
    return iq.astype(np.complex64)


def _run(ctx_factory, iq, freqs, exact=False, chunks=None):
    ctx = ctx_factory()
    try:
        ctx.set_exact(exact)
        rx = ctx.receiver_open(FS, BLK, 0)
        chs = [ctx.channel_open(rx, f, "FT8") for f in freqs]
        ctx.slot_boundary("FT8", 100)
        if chunks is None:
            ctx.push_iq(rx, iq)
        else:
            pos = 0
            k = 0
            while pos < len(iq):
                n = min(chunks[k % len(chunks)] * BLK, len(iq) - pos)
                ctx.push_iq(rx, iq[pos:pos + n]); pos += n; k += 1
        ctx.slot_boundary("FT8", 115)
        return [(ctx.fetch_audio_f32(c)[0], ctx.fetch_frame(c)) for c in chs]
    finally:
        ctx.close()


@pytest.fixture
def ctx_factory():
    import cwsl_digi_amd as P
    return lambda: P.Context(0)


def test_power_of_two_scaling_is_exact_at_full_slot_size(ctx_factory):
    """Every operation of the path is a float multiply/add: scaling the IQ by 4 scales the float audio by exactly 4, bit for
    bit, in both modes -- 180 000 outputs x 3 channels, recursive phasor included."""
    iq = _slot_iq(1)
    freqs = [-26000, 41000, -77000]
    for exact in (False, True):
        a = _run(ctx_factory, iq, freqs, exact)
        b = _run(ctx_factory, iq * np.float32(4.0), freqs, exact)
        for (fa, _), (fb, _) in zip(a, b):
            assert np.array_equal(fa * np.float32(4.0), fb)
            assert np.abs(fa).max() > 1000.0


def test_push_granularity_does_not_change_a_bit(ctx_factory):
    """One 15 s push, single blocks, and ragged multi-block pushes (ring wraps many times) give identical frames."""
    iq = _slot_iq(2)
    freqs = [-26000, 41000]
    ref = _run(ctx_factory, iq, freqs)
    for chunks in ([1], [7, 1, 64, 3, 200]):
        got = _run(ctx_factory, iq, freqs, chunks=chunks)
        for (fa, ga), (fb, gb) in zip(ref, got):
            assert np.array_equal(fa, fb) and np.array_equal(ga["i16"], gb["i16"]) and ga["n_valid"] == gb["n_valid"] == SLOT // 16


def test_channels_on_one_receiver_are_independent(ctx_factory):
    iq = _slot_iq(3)
    alone = _run(ctx_factory, iq, [41000])[0]
    crowd = _run(ctx_factory, iq, [-90000, -26000, 41000, 1234, 87000, -50000, 41000, 60000])
    assert np.array_equal(alone[0], crowd[2][0]) and np.array_equal(alone[1]["i16"], crowd[2][1]["i16"])
    assert np.array_equal(crowd[2][0], crowd[6][0])                   # the same tuning twice: the same frame


def test_fast_mode_within_tolerance_of_exact_mode_at_full_size(ctx_factory):
    """tolerance 1e-5 of frame peak (north star); measured ~4e-7.  int16 may differ by 1 LSB at rounding ties."""
    iq = _slot_iq(4)
    freqs = [-26000, 41000, -77000]
    fast = _run(ctx_factory, iq, freqs, exact=False)
    exact = _run(ctx_factory, iq, freqs, exact=True)
    for (fa, ga), (fb, gb) in zip(fast, exact):
        peak = float(np.abs(fb).max())
        assert float(np.abs(fa.astype(np.float64) - fb).max()) <= 1e-5 * peak
        assert int(np.abs(ga["i16"].astype(np.int32) - gb["i16"]).max()) <= 1
