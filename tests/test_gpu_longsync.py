"""GPU: candidate search of the 120 s modes (BASELINE.json configs[4]) through the C ABI vs the
repository's restatement (oracle/longsync_oracle.c).  PARITY UNPINNED by the reference (wsprd / jt9 -W are not vendored);
against the restatement every stage is BIT-EXACT: WSPR's 375 Hz baseband, its 359 x 512 spectra, the normalised smoothed
spectrum and the candidate list with the coarse (frequency, shift, drift, sync) estimates; FST4W's band of the long
transform, the normalised comb spectrum and the CLEAN list.  The restatement is fed the GPU's int16 frame here (stage parity);
the chain from IQ is covered at full size by tests/test_gpu_fullsize.py::test_config4 and scripts/run_configs.py."""
import numpy as np
import pytest

from longsync_signal import wspr_iq, fst4w_iq

pytestmark = pytest.mark.gpu
FS, BLK = 192000, 2048
N = 23040000 // BLK * BLK


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def long_frames(oracle):
    """One receiver, a WSPR and an FST4W-120 decoder on it; 120 s of noise + two transmissions each."""
    import cwsl_digi_amd as P
    rng = np.random.default_rng(5)
    f_w, f_f = 20000, -45000
    iq = oracle.synth_iq(77, N, FS)                                    # noise
    iq = iq + wspr_iq(FS, N, f_w, 1500 + 41.0, 2.0, 600.0, rng) + wspr_iq(FS, N, f_w, 1500 - 77.0, 1.2, 250.0, rng, drift_hz=-2.0)
    iq = iq + fst4w_iq(FS, N, f_f, 1500.0, 1.0, 500.0, rng) + fst4w_iq(FS, N, f_f, 1451.0, 0.4, 250.0, rng)
    iq = iq.astype(np.complex64)
    with P.Context(0) as ctx:
        ctx.enable_long_sync(True)
        rx = ctx.receiver_open(FS, BLK, 0)
        cw = ctx.channel_open(rx, f_w, "WSPR")
        cf = ctx.channel_open(rx, f_f, "FST4W-120")
        ctx.slot_boundary("S120", 120)
        assert ctx.fetch_wspr_candidates(cw) is None                    # nothing finalised yet
        for k in range(0, N, 256 * BLK):
            ctx.push_iq(rx, iq[k:k + 256 * BLK])
        ctx.slot_boundary("S120", 240)
        out = dict(
            wspr_frame=ctx.fetch_frame(cw)["i16"].copy(), fst_frame=ctx.fetch_frame(cf)["i16"].copy(),
            wspr_cands=ctx.fetch_wspr_candidates(cw), fst_cands=ctx.fetch_fst4w_candidates(cf),
            iq=ctx.long_sync_debug(cw, "iq"), ps=ctx.long_sync_debug(cw, "ps"), smspec=ctx.long_sync_debug(cw, "smspec"),
            s2=ctx.long_sync_debug(cf, "s2"), band=ctx.long_sync_debug(cf, "band"),
            wspr_slot=ctx.fetch_slot(cw), fst_slot=ctx.fetch_slot(cf),
            wspr_epoch=ctx.fetch_wspr_candidates(cw, with_epoch=True)[1], fst_epoch=ctx.fetch_fst4w_candidates(cf, with_epoch=True)[1])
        with pytest.raises(P.CwslGpuError):
            ctx.fetch_fst4w_candidates(cw)                               # a WSPR channel has no FST4W list
    return out


def test_wspr_stages_and_candidates_bit_exact(oracle, long_frames):
    g = long_frames
    ref, arr = oracle.wspr_search(g["wspr_frame"], want_arrays=True)
    assert np.array_equal(_bits(g["iq"].real), _bits(arr["idat"])) and np.array_equal(_bits(g["iq"].imag), _bits(arr["qdat"]))
    assert np.array_equal(_bits(g["ps"]), _bits(arr["ps"]))
    assert np.array_equal(_bits(g["smspec"]), _bits(arr["smspec"]))
    got = g["wspr_cands"]
    assert len(got) == len(ref) >= 2
    for a, b in zip(got, ref):
        assert [np.float32(x).view(np.uint32) for x in a[:4]] == [np.float32(x).view(np.uint32) for x in b[:4]] and a[4] == b[4]
    # and the transmissions are where they were put (centre of the four tones = tone 0 + 1.5 spacings)
    for f, t0 in ((41.0 + 2.197, 2.0), (-77.0 + 2.197, 1.2)):
        hit = [c for c in got if abs(c[0] - f) <= 1.5]
        assert hit and hit[0][3] > 0.2 and abs(hit[0][4] / 375.0 - t0) <= 0.4


def test_fst4w_stages_and_candidates_bit_exact(oracle, long_frames):
    g = long_frames
    ref, arr = oracle.fst4w_candidates(g["fst_frame"], want_arrays=True)
    nband = len(g["band"])
    power = g["band"].real.astype(np.float32) ** 2 + g["band"].imag.astype(np.float32) ** 2       # float32, un-fused: as the oracle's band_o
    assert np.array_equal(_bits(power), _bits(arr["band"][:nband]))
    n = min(len(g["s2"]), len(arr["s2"]))
    assert np.array_equal(_bits(g["s2"][:n]), _bits(arr["s2"][:n]))
    got = g["fst_cands"]
    assert len(got) == len(ref) >= 2
    for a, b in zip(got, ref):
        assert np.float32(a[0]).view(np.uint32) == np.float32(b[0]).view(np.uint32)
        assert np.float32(a[1]).view(np.uint32) == np.float32(b[1]).view(np.uint32) and a[2] == b[2]
    baud = 12000.0 / 8200.0
    assert abs(got[0][0] - (1500.0 + 1.5 * baud)) <= baud
    assert any(abs(c[0] - (1451.0 + 1.5 * baud)) <= baud for c in got[:5])


def test_fetch_slot_carries_the_long_modes_lists(long_frames):
    """ABI 5: cwslg_fetch_slot on WSPR / FST4W-120 channels -- the frame, its start epoch and the list in the channel's own record type, under one
    ticket; the list fetches name the same epoch."""
    g = long_frames
    for slot, frame, cands, kind, epoch in ((g["wspr_slot"], g["wspr_frame"], g["wspr_cands"], "WSPR", g["wspr_epoch"]),
                                            (g["fst_slot"], g["fst_frame"], g["fst_cands"], "FST4W", g["fst_epoch"])):
        assert slot["t_start"] == 120 == epoch and slot["list_kind"] == kind and slot["n_valid"] == N // 16
        assert np.array_equal(slot["i16"], frame)
        assert [tuple(np.float32(x).view(np.uint32) if isinstance(x, float) else x for x in c) for c in slot["list"]] == \
               [tuple(np.float32(x).view(np.uint32) if isinstance(x, float) else x for x in c) for c in cands]
        assert len(cands) >= 2 and slot["ft4_sync"] == []


def test_short_slot_zero_tail(oracle):
    """A 120 s slot that ended after 50 s: the frame's zero tail goes through both searches (wsprd reads 114 s, jt9 120 s of it);
    lists still bit-identical to the restatement."""
    import cwsl_digi_amd as P
    n = 9600000 // BLK * BLK                                            # 50 s
    rng = np.random.default_rng(8)
    f_w, f_f = -30000, 55000
    iq = oracle.synth_iq(78, n, FS)
    iq = iq + wspr_iq(FS, n, f_w, 1500 - 20.0, 1.5, 700.0, rng) + fst4w_iq(FS, n, f_f, 1530.0, 0.7, 600.0, rng)
    iq = iq.astype(np.complex64)
    with P.Context(0) as ctx:
        ctx.enable_long_sync(True)
        rx = ctx.receiver_open(FS, BLK, 0)
        cw, cf = ctx.channel_open(rx, f_w, "WSPR"), ctx.channel_open(rx, f_f, "FST4W-120")
        ctx.slot_boundary("S120", 120)
        for k in range(0, n, 256 * BLK):
            ctx.push_iq(rx, iq[k:k + 256 * BLK])
        ctx.slot_boundary("S120", 240)
        fw, ff = ctx.fetch_frame(cw), ctx.fetch_frame(cf)
        assert fw["n_valid"] == n // 16 and not fw["i16"][n // 16:].any()
        gw, gf = ctx.fetch_wspr_candidates(cw), ctx.fetch_fst4w_candidates(cf)
    rw, rf = oracle.wspr_search(fw["i16"]), oracle.fst4w_candidates(ff["i16"])
    b = lambda t: [np.float32(x).view(np.uint32) if isinstance(x, float) else x for x in t]
    assert [b(t) for t in gw] == [b(t) for t in rw] and len(rw) >= 1
    assert [b(t) for t in gf] == [b(t) for t in rf] and len(rf) >= 1
    assert any(abs(c[0] - (-20.0 + 2.197)) <= 1.5 for c in gw)
