"""GPU: FT8 sync stage through the C ABI vs the repository's CPU restatement (oracle/sync_oracle.c).
PARITY UNPINNED by the reference (it has no sync code); against the restatement everything is BIT-EXACT:
symbol spectra, per-bin peaks and lags, and the final candidate list."""
import numpy as np
import pytest

from ft8_signal import ft8_iq

pytestmark = pytest.mark.gpu
FS, BLK = 192000, 2048


def _run(ctx, oracle, f, specs, seed, maxcand=200, syncmin=1.5, lo=200, hi=3000, n_iq=None):
    n = n_iq or (2880000 // BLK * BLK)
    rng = np.random.default_rng(seed)
    iq = oracle.synth_iq(seed, n, FS)
    for audio_hz, t0, amp in specs:
        iq = iq + ft8_iq(FS, n, f, audio_hz, t0, amp, rng)
    iq = iq.astype(np.complex64)
    ctx.enable_sync(True, syncmin, maxcand, lo, hi)
    rx = ctx.receiver_open(FS, BLK, 0)
    ch = ctx.channel_open(rx, f, "FT8")
    ctx.slot_boundary("FT8", 1)
    for k in range(0, n, 128 * BLK):
        ctx.push_iq(rx, iq[k:k + 128 * BLK])
    ctx.slot_boundary("FT8", 16)
    fr = ctx.fetch_frame(ch)["i16"]
    return ch, fr


def test_sync_bit_exact_vs_restatement(ctx, oracle):
    specs = [(700.0, 0.5, 3000.0), (1531.25, 1.3, 2000.0), (2400.0, 0.1, 1500.0), (2403.0, 0.12, 900.0)]
    ch, fr = _run(ctx, oracle, 10000, specs, 5)
    # the stage runs on the GPU's own int16 frame; feed the SAME frame to the restatement
    cands_ref, arr = oracle.ft8_sync(fr, 200, 3000, 1.5, 200, want_arrays=True)
    s_gpu = ctx.sync_debug(ch, "spectra")
    s_ref = oracle.ft8_spectra(fr, s_gpu.shape[1])
    assert np.array_equal(s_gpu.view(np.uint32), s_ref.view(np.uint32))
    ia, ib = 64, 960
    for name in ("red", "red2"):
        g = ctx.sync_debug(ch, name)
        assert np.array_equal(g[ia:ib + 1].view(np.uint32), arr[name][ia:ib + 1].view(np.uint32)), name
    for name in ("jpeak", "jpeak2"):
        assert np.array_equal(ctx.sync_debug(ch, name)[ia:ib + 1], arr[name][ia:ib + 1]), name
    cands = ctx.fetch_candidates(ch, 200)
    assert len(cands) == len(cands_ref) >= 3
    for a, b in zip(cands, cands_ref):
        assert a[0] == b[0] and a[1] == b[1]
        assert np.float32(a[2]).view(np.uint32) == np.float32(b[2]).view(np.uint32)
        assert np.float32(a[3]) == np.float32(b[3]) and np.float32(a[4]) == np.float32(b[4])


@pytest.mark.parametrize("maxcand,syncmin,lo,hi", [(600, 1.2, 100, 4000), (10, 1.5, 200, 3000), (50, 2.0, 500, 2500)])
def test_sync_parameter_sweep(ctx, oracle, maxcand, syncmin, lo, hi):
    specs = [(800.0 + 37 * k, 0.2 + 0.11 * k, 2500.0 - 150 * k) for k in range(9)]
    ch, fr = _run(ctx, oracle, -30000, specs, 11, maxcand, syncmin, lo, hi)
    ref = oracle.ft8_sync(fr, lo, hi, syncmin, maxcand)
    got = ctx.fetch_candidates(ch, 600)
    assert [(c[0], c[1], np.float32(c[2]).view(np.uint32)) for c in got] == \
           [(c[0], c[1], np.float32(c[2]).view(np.uint32)) for c in ref]
    assert len(got) <= maxcand


@pytest.mark.parametrize("hi", [2900, 2960, 2962, 3037, 3062, 3100, 5900])
def test_spectra_rows_at_every_form_of_the_last_stage(ctx, oracle, hi):
    """The spectra kernel's last stage has three forms by the stored row length: rows below 960 bins (plain items), 960..992 (the bins above 960
    packed into row 0's idle lanes, the self-paired bins on lane 0 -- FT8's default search), and wider rows (four bins per item): every stored bin of
    every symbol step is the restatement's, bit for bit, at row lengths on both sides of each switch."""
    specs = [(600.0 + 310 * k, 0.2 + 0.13 * k, 2500.0 - 200 * k) for k in range(8)]
    ch, fr = _run(ctx, oracle, 20000, specs, 17, 200, 1.5, 200, hi)
    s_gpu = ctx.sync_debug(ch, "spectra")
    s_ref = oracle.ft8_spectra(fr, s_gpu.shape[1])
    assert s_gpu.shape[0] == 372 and np.array_equal(s_gpu.view(np.uint32), s_ref.view(np.uint32)), s_gpu.shape
    ref = oracle.ft8_sync(fr, 200, hi, 1.5, 200)
    got = ctx.fetch_candidates(ch, 200)
    assert [(c[0], c[1], np.float32(c[2]).view(np.uint32)) for c in got] == [(c[0], c[1], np.float32(c[2]).view(np.uint32)) for c in ref]


def test_sync_short_frame_zero_tail(ctx, oracle):
    """A slot that ended early: the zero tail gives all-zero symbol windows (0/0 defined as 0 in both)."""
    specs = [(1200.0, 0.5, 3000.0)]
    ch, fr = _run(ctx, oracle, 5000, specs, 3, n_iq=900 * BLK)
    ref = oracle.ft8_sync(fr, 200, 3000, 1.5, 200)
    got = ctx.fetch_candidates(ch, 200)
    assert [(c[0], c[1], np.float32(c[2]).view(np.uint32)) for c in got] == \
           [(c[0], c[1], np.float32(c[2]).view(np.uint32)) for c in ref]


def test_sync_many_channels_one_launch(ctx, oracle):
    """16 FT8 channels + 2 FT4 channels (no FT8 sync for those) finalised by one boundary."""
    n = 2880000 // BLK * BLK
    rng = np.random.default_rng(2)
    freqs = [-80000 + 9000 * k for k in range(16)]
    iq = oracle.synth_iq(99, n, FS)
    for k, f in enumerate(freqs):
        iq = iq + ft8_iq(FS, n, f, 500.0 + 140 * k, 0.3 + 0.05 * k, 2500.0, rng)
    iq = iq.astype(np.complex64)
    ctx.enable_sync(True, 1.5, 100, 200, 3000)
    rx = ctx.receiver_open(FS, BLK, 0)
    chans = [ctx.channel_open(rx, f, "FT8") for f in freqs]
    ctx.slot_boundary("FT8", 1)
    for k in range(0, n, 128 * BLK):
        ctx.push_iq(rx, iq[k:k + 128 * BLK])
    ctx.slot_boundary("FT8", 16)
    for k, ch in enumerate(chans):
        fr = ctx.fetch_frame(ch)["i16"]
        ref = oracle.ft8_sync(fr, 200, 3000, 1.5, 100)
        got = ctx.fetch_candidates(ch, 100)
        assert [(c[0], c[1], np.float32(c[2]).view(np.uint32)) for c in got] == \
               [(c[0], c[1], np.float32(c[2]).view(np.uint32)) for c in ref]
        want_bin = int(round((500.0 + 140 * k) / 3.125))
        assert any(abs(c[0] - want_bin) <= 1 for c in got)


@pytest.mark.parametrize("maxcand,syncmin,lo,hi", [(600, 1.2, 100, 5800), (200, 1.5, 200, 3000), (50, 2.0, 437, 2313), (100, 1.5, 0, 1000)])
def test_sync_per_channel_form(ctx, oracle, maxcand, syncmin, lo, hi):
    """With at least two workgroups' worth of FT8 channels per CU a boundary runs ft8_sync_chan_kernel (one workgroup per channel: sliding
    LDS window over all bands + fused candidate selection) instead of one workgroup per band: 520 channels share one receiver here; the
    per-bin peaks and lags and the candidate lists of channels spread over the range must equal the restatement's bit for bit -- for the
    default search, the widest one (row pitch 1952 bins: the spectra kernel's generic upper half) and a range whose edges fall inside bands."""
    fs, blk = 48000, 2048
    n = 720000 // blk * blk
    rng = np.random.default_rng(5)
    freqs = [int(f) for f in rng.integers(-fs // 2, fs // 2 - 6500, 520)]
    iq = oracle.synth_iq(77, n, fs)
    probe = [0, 1, 257, 519]
    for k in probe:
        for j in range(3):
            iq = iq + ft8_iq(fs, n, freqs[k], 400.0 + 700.0 * j + 13.0 * k % 97, 0.2 + 0.3 * j, 1500.0 + 400.0 * j, rng)
    iq = iq.astype(np.complex64)
    ctx.enable_sync(True, syncmin, maxcand, lo, hi)
    rx = ctx.receiver_open(fs, blk, 0)
    chans = [ctx.channel_open(rx, f, "FT8") for f in freqs]
    ctx.slot_boundary("FT8", 1)
    for k in range(0, n, 64 * blk):
        ctx.push_iq(rx, iq[k:k + 64 * blk])
    ctx.slot_boundary("FT8", 16)
    for k in probe:
        ch = chans[k]
        fr = ctx.fetch_frame(ch)["i16"]
        ref, arr = oracle.ft8_sync(fr, lo, hi, syncmin, maxcand, want_arrays=True)
        ia, ib = max(1, int(round(lo / 3.125))), min(int(round(hi / 3.125)), 1920 - 12)      # sync8: ia = nint(nfa / df), ib = nint(nfb / df)
        for name in ("red", "red2"):
            assert np.array_equal(ctx.sync_debug(ch, name)[ia:ib + 1].view(np.uint32), arr[name][ia:ib + 1].view(np.uint32)), (k, name)
        for name in ("jpeak", "jpeak2"):
            assert np.array_equal(ctx.sync_debug(ch, name)[ia:ib + 1], arr[name][ia:ib + 1]), (k, name)
        got = ctx.fetch_candidates(ch, 600)
        assert [(c[0], c[1], np.float32(c[2]).view(np.uint32)) for c in got] == \
               [(c[0], c[1], np.float32(c[2]).view(np.uint32)) for c in ref], k
        assert len(got) <= maxcand


# ---------------------------------------------------------------------------------------------- FT4
from ft8_signal import ft4_iq


def _run_ft4(ctx, oracle, f, specs, seed, maxcand=200, lo=200, hi=4000, syncmin4=1.2):
    n = 1440000 // BLK * BLK
    rng = np.random.default_rng(seed)
    iq = oracle.synth_iq(seed, n, FS)
    for audio_hz, t0, amp in specs:
        iq = iq + ft4_iq(FS, n, f, audio_hz, t0, amp, rng)
    iq = iq.astype(np.complex64)
    ctx.enable_sync(True, 1.5, maxcand, lo, hi)
    ctx.set_ft4_syncmin(syncmin4)
    rx = ctx.receiver_open(FS, BLK, 0)
    ch = ctx.channel_open(rx, f, "FT4")
    ctx.slot_boundary("FT4", 1)
    for k in range(0, n, 128 * BLK):
        ctx.push_iq(rx, iq[k:k + 128 * BLK])
    ctx.slot_boundary("FT4", 8)
    return ch, ctx.fetch_frame(ch)["i16"]


def _key(cands):
    return [(c[0], np.float32(c[2]).view(np.uint32), np.float32(c[3]).view(np.uint32)) for c in cands]


def test_ft4_bit_exact_vs_restatement(ctx, oracle):
    specs = [(700.0, 0.3, 2500.0), (1800.0, 0.6, 1500.0), (3100.0, 0.2, 2000.0), (3150.0, 1.0, 800.0)]
    ch, fr = _run_ft4(ctx, oracle, -40000, specs, 9)
    ref, arr = oracle.ft4_candidates(fr, 200.0, 4000.0, 1.2, 200, want_arrays=True)
    s_gpu = ctx.sync_debug(ch, "spectra")
    assert s_gpu.shape == (122, 1168)
    s_ref = oracle.ft4_spectra(fr)
    assert np.array_equal(s_gpu[:, :1153].view(np.uint32), s_ref.view(np.uint32))
    assert np.array_equal(ctx.sync_debug(ch, "red2")[:1153].view(np.uint32), arr["sbase"].view(np.uint32))     # baseline
    assert np.array_equal(ctx.sync_debug(ch, "red")[:1153].view(np.uint32), arr["savsm"].view(np.uint32))      # savsm/sbase
    got = ctx.fetch_candidates(ch, 200)
    assert len(got) == len(ref) >= 3 and _key(got) == _key(ref)
    assert all(c[1] == 0 and c[4] == 0.0 for c in got)


@pytest.mark.parametrize("maxcand,lo,hi,smin", [(5, 200, 4000, 1.2), (600, 100, 5000, 1.05), (50, 500, 2500, 2.0)])
def test_ft4_parameter_sweep(ctx, oracle, maxcand, lo, hi, smin):
    specs = [(600.0 + 290 * k, 0.1 + 0.1 * k, 2600.0 - 200 * k) for k in range(8)]
    ch, fr = _run_ft4(ctx, oracle, 20000, specs, 21, maxcand, lo, hi, smin)
    ref = oracle.ft4_candidates(fr, float(lo), float(hi), smin, maxcand)
    got = ctx.fetch_candidates(ch, 600)
    assert _key(got) == _key(ref) and len(got) <= maxcand


def test_mixed_ft8_ft4_sync_in_one_context(ctx, oracle):
    """BASELINE configs[2]: FT8 and FT4 slots side by side, each with its own sync stage."""
    n8 = 2880000 // BLK * BLK
    rng = np.random.default_rng(4)
    iq = oracle.synth_iq(31, n8, FS)
    iq = iq + ft8_iq(FS, n8, 10000, 1200.0, 0.5, 2500.0, rng) + ft4_iq(FS, n8, -30000, 900.0, 0.3, 2500.0, rng) \
        + ft4_iq(FS, n8, -30000, 2100.0, 7.8, 2000.0, rng)
    iq = iq.astype(np.complex64)
    ctx.enable_sync(True, 1.5, 100, 200, 3000)
    rx = ctx.receiver_open(FS, BLK, 0)
    c8 = ctx.channel_open(rx, 10000, "FT8")
    c4 = ctx.channel_open(rx, -30000, "FT4")
    ctx.slot_boundary("FT8", 1); ctx.slot_boundary("FT4", 1)
    half = n8 // 2 // BLK * BLK
    ctx.push_iq(rx, iq[:half]); ctx.slot_boundary("FT4", 8)
    fr4a = ctx.fetch_frame(c4)["i16"]; got4a = ctx.fetch_candidates(c4, 100)
    ctx.push_iq(rx, iq[half:]); ctx.slot_boundary("FT4", 15); ctx.slot_boundary("FT8", 15)
    fr4b = ctx.fetch_frame(c4)["i16"]; got4b = ctx.fetch_candidates(c4, 100)
    fr8 = ctx.fetch_frame(c8)["i16"]; got8 = ctx.fetch_candidates(c8, 100)
    assert _key(got4a) == _key(oracle.ft4_candidates(fr4a, 200.0, 3000.0, 1.2, 100))
    assert _key(got4b) == _key(oracle.ft4_candidates(fr4b, 200.0, 3000.0, 1.2, 100))
    ref8 = oracle.ft8_sync(fr8, 200, 3000, 1.5, 100)
    assert [(c[0], c[1], np.float32(c[2]).view(np.uint32)) for c in got8] == [(c[0], c[1], np.float32(c[2]).view(np.uint32)) for c in ref8]
    assert any(abs(c[3] - 900.0) <= 40 for c in got4a[:4]) and any(abs(c[3] - 2100.0) <= 40 for c in got4b[:4])
    assert any(abs(c[3] - 1200.0) <= 4 for c in got8[:3])


def test_candidate_order_option_and_dense_lists(ctx, oracle):
    """cwslg_set_candidate_order (ABI 5) and the near-duplicate rule on dense lists.  One receiver, an FT8 channel carrying 16 signals and an FT4
    channel carrying 6; the same slot is searched in both orders with a list that is cut (maxcand 40) and one that is not (600): every list
    equals the restatement's bit for bit, 'freq' keeps the lowest 40 in frequency where 'sync' keeps the strongest 40, and uncut they hold the
    same entries.  With 16 signals the FT8 list has hundreds of entries whose survival is decided by the float32 `tdiff < 0.04` test
    (tests/test_sync_oracle.py::test_tdiff_boundary_decides_real_lists shows the alternatives differ by dozens of entries on such frames)."""
    n8 = 2880000 // BLK * BLK
    rng = np.random.default_rng(77)
    iq = oracle.synth_iq(77, n8, FS)
    for _ in range(16):
        iq = iq + ft8_iq(FS, n8, 10000, rng.uniform(250, 2900), rng.uniform(0.1, 1.9), rng.uniform(600, 2500), rng)
    for _ in range(6):
        iq = iq + ft4_iq(FS, n8, -30000, rng.uniform(300, 2800), rng.uniform(0.2, 1.0), rng.uniform(800, 2500), rng)
    iq = iq.astype(np.complex64)
    rx = ctx.receiver_open(FS, BLK, 0)
    c8 = ctx.channel_open(rx, 10000, "FT8")
    c4 = ctx.channel_open(rx, -30000, "FT4")
    key = lambda lst: [(c[0], c[1], np.float32(c[2]).view(np.uint32), np.float32(c[3]).view(np.uint32)) for c in lst]
    lists = {}
    epoch = 100
    for order in ("sync", "freq"):
        for maxcand in (40, 600):
            ctx.enable_sync(True, 1.2, maxcand, 200, 3000)
            ctx.set_candidate_order(order)
            ctx.slot_boundary("FT8", epoch); ctx.slot_boundary("FT4", epoch)
            for k in range(0, n8, 128 * BLK):
                ctx.push_iq(rx, iq[k:k + 128 * BLK])
            epoch += 15
            ctx.slot_boundary("FT8", epoch); ctx.slot_boundary("FT4", epoch)
            fr8, fr4 = ctx.fetch_frame(c8)["i16"], ctx.fetch_frame(c4)["i16"]
            g8, g4 = ctx.fetch_candidates(c8, 600), ctx.fetch_candidates(c4, 600)
            assert key(g8) == key(oracle.ft8_sync(fr8, 200, 3000, 1.2, maxcand, order=order)), (order, maxcand)
            assert key(g4) == key(oracle.ft4_candidates(fr4, 200.0, 3000.0, 1.2, maxcand, order=order)), (order, maxcand)
            lists[(order, maxcand)] = (g8, g4)
            epoch += 15
    s40, s600 = lists[("sync", 40)][0], lists[("sync", 600)][0]
    f40, f600 = lists[("freq", 40)][0], lists[("freq", 600)][0]
    assert len(s600) > 100 and len(s40) == len(f40) == 40
    assert sorted(key(s600)) == sorted(key(f600))                                       # uncut: the same entries
    assert key(s40) == key(s600[:40]) and key(f40) == key(f600[:40])
    assert [c[0] for c in f600] == sorted(c[0] for c in f600) and [c[2] for c in s600] == sorted((c[2] for c in s600), reverse=True)
    assert {c[:2] for c in s40} != {c[:2] for c in f40}                                 # cut: different entries
    g4s, g4f = lists[("sync", 600)][1], lists[("freq", 600)][1]
    assert sorted(key(g4s)) == sorted(key(g4f)) and [c[0] for c in g4f] == sorted(c[0] for c in g4f) and len(g4f) >= 4


def test_fused_finalise_rewrites_the_tail_only_where_needed(ctx, oracle):
    """Round 6: for FT8 channels with the sync stage on the slot's finalise (prepareAudio + int16, Instance.cpp:294-338, 238-241) runs inside
    symbol_spectra_v2_kernel, which converts what its windows cover (samples 0 .. 179999) and rewrites the 20 s frame's tail beyond them only
    as far as this slot or the previous one put samples there.  A late boundary (205 000 samples: tail in use), then a short slot (the old tail
    must become zeros again), then an ordinary one, then one without the sync stage (the stand-alone finalise takes over) and with it again:
    every int16 frame equals the oracle's whole, zero tail included, and the lists follow."""
    f = 10000
    lens = [1602 * BLK, 400 * BLK, 1405 * BLK, 1300 * BLK, 1406 * BLK]              # 205 056 / 51 200 / 179 840 / 166 400 / 179 968 outputs
    rng = np.random.default_rng(9)
    iq = oracle.synth_iq(91, sum(lens), FS).astype(np.complex64)
    iq = (iq + ft8_iq(FS, len(iq), f, 1500.0, 0.7, 2500.0, rng) + ft8_iq(FS, len(iq), f, 900.0, 17.5 + 4.3 + 15.0, 2000.0, rng)).astype(np.complex64)
    ctx.enable_sync(True, 1.5, 100, 200, 3000)
    rx = ctx.receiver_open(FS, BLK, 0)
    ch = ctx.channel_open(rx, f, "FT8")
    oc = oracle.Channel("FT8", FS, BLK, f)
    ctx.slot_boundary("FT8", 10); assert oc.boundary(10) is None
    pos = 0
    for k, n in enumerate(lens):
        if k == 3:
            ctx.enable_sync(False)
        if k == 4:
            ctx.enable_sync(True, 1.5, 100, 200, 3000)
        for q in range(pos, pos + n, 128 * BLK):
            ctx.push_iq(rx, iq[q:min(q + 128 * BLK, pos + n)])
        oc.push_many(iq[pos:pos + n]); pos += n
        ctx.slot_boundary("FT8", 25 + 15 * k)
        ref = oc.boundary(25 + 15 * k, want_f32=True)
        g = ctx.fetch_frame(ch)
        assert g["n_valid"] == n // 16
        if ctx.mode == "exact":
            assert np.array_equal(g["i16"], ref["i16"]), k
        else:
            assert np.abs(g["i16"].astype(np.int32) - ref["i16"].astype(np.int32)).max() <= 1, k
        assert not g["i16"][g["n_valid"]:].any(), k                                 # the reference's zero tail, whatever the buffer held before
        if k != 3:
            got = ctx.fetch_candidates(ch, 100)
            want = oracle.ft8_sync(g["i16"], 200, 3000, 1.5, 100)
            assert [(c[0], c[1], np.float32(c[2]).view(np.uint32)) for c in got] == [(c[0], c[1], np.float32(c[2]).view(np.uint32)) for c in want], k
    st = ctx.stats()
    assert st["finalize_launches"] == 2            # the discarded first boundary and the slot without the sync stage; the other four rode in the spectra kernel
