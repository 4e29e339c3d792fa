"""CPU: the restatement of the candidate searches (oracle/sync_oracle.c, longsync_oracle.c -- what the kernels are bit-identical to) against an
independent numpy implementation of the same published selection rules (tests/indep_sync.py) on 32 synthetic frames per mode.  The two differ in
precision and FFT, so a decision that sits within 1e-3 of a threshold, of the percentile cut or of a second maximum may fall either way; everything
else -- which bins, which lags, which order -- must agree."""
import numpy as np
import pytest

import indep_sync as I
from ft8_signal import ft8_iq, ft4_audio
from longsync_signal import wspr_audio, fst4w_audio, to_i16

N_FRAMES = 32


def _ft8_frame(seed, n_sig):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(240000) * 300.0
    for _ in range(n_sig):
        a += ft8_iq(12000, 240000, 0.0, rng.uniform(250, 2900), rng.uniform(0.1, 1.9), rng.uniform(150, 1500), rng).real
    return to_i16(a)


def test_ft8_selection_agrees_with_the_independent_implementation(oracle):
    total = odd = 0
    for k in range(N_FRAMES):
        fr = _ft8_frame(1000 + k, n_sig=3 + k % 9)
        got, arr = oracle.ft8_sync(fr, 200, 3000, 1.5, 600, want_arrays=True)
        want, wa = I.ft8_candidates(fr, 200, 3000, 1.5, 600)
        # the per-bin arrays first: normalised sync and its lag, both windows
        b = wa["bins"]
        for name in ("red", "red2"):                                # (the restatement hands back the metric before its percentile normalisation)
            raw = arr[name][b] / np.sort(arr[name][b])[int(np.floor(0.40 * len(b) + 0.5)) - 1]
            assert np.allclose(raw, wa[name], rtol=2e-4), (k, name)
        near_tie = 0
        for name in ("jpeak", "jpeak2"):
            bad = np.nonzero(arr[name][b] != wa[name])[0]
            near_tie += len(bad)                                    # a second maximum within 2e-4 may win in one precision and lose in the other
            assert len(bad) <= 2, (k, name, bad)
        gk = {(c[0], c[1]): c[2] for c in got}
        wk = {(c[0], c[1]): c[2] for c in want}
        total += len(wk)
        for key in set(gk) ^ set(wk):
            v = gk.get(key, wk.get(key))
            assert abs(v - 1.5) <= 3e-3 or near_tie, (k, key, v)     # only a candidate AT the threshold (or a lag tie) may differ
            odd += 1
        # ORDER -- an open point this comparison surfaced: the restatement (and the kernel) hand the list out strongest first; this module, from
        # its reading of sync8.f90 ("Sort by frequency", the "Sort by sync" lines commented out), in ascending frequency.  Upstream's source is
        # not here to settle it; the two orders hold the same entries whenever the list is not cut at maxcand (checked here as sets, and the
        # restatement's own order below), and the decoder behind the list tries every entry.
        assert [c[2] for c in got] == sorted((c[2] for c in got), reverse=True)
        for key in set(gk) & set(wk):
            assert gk[key] == pytest.approx(wk[key], rel=3e-4)
    assert total > 40 * N_FRAMES and odd <= 4, (total, odd)


def test_ft8_cut_at_maxcand_is_where_the_two_orders_differ(oracle):
    """With more candidates than maxcand the restatement keeps the STRONGEST maxcand, the frequency-ordered reading the LOWEST in frequency: the
    one place the open ordering point changes which entries exist.  The synthetic soak and the north-star workload stay below maxcand.  Pinned
    here so that a change of either side is seen."""
    fr = _ft8_frame(77, n_sig=14)
    full = oracle.ft8_sync(fr, 200, 3000, 1.2, 600)
    got = oracle.ft8_sync(fr, 200, 3000, 1.2, 40)
    want, _ = I.ft8_candidates(fr, 200, 3000, 1.2, 40)
    assert len(full) > 40 and len(got) == len(want) == 40
    assert [c[:2] for c in got] == [c[:2] for c in full[:40]]                         # the strongest 40 of the full list
    allk, _ = I.ft8_candidates(fr, 200, 3000, 1.2, 600)
    assert [c[:2] for c in want] == [c[:2] for c in allk[:40]]                        # the 40 lowest in frequency of the same full set
    assert {c[:2] for c in full} == {c[:2] for c in allk}


def _ft4_frame(seed, n_sig):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(150000) * 300.0
    for _ in range(n_sig):
        a += ft4_audio(150000, rng.uniform(300, 3800), rng.uniform(0.2, 1.0), rng.uniform(200, 1500), rng)
    return to_i16(a)


def test_ft4_selection_agrees_with_the_independent_implementation(oracle):
    total = 0
    for k in range(N_FRAMES):
        fr = _ft4_frame(2000 + k, n_sig=2 + k % 7)
        got, arr = oracle.ft4_candidates(fr, 200.0, 4000.0, 1.2, 200, want_arrays=True)
        want, savsm, sbase = I.ft4_candidates(fr, 200.0, 4000.0, 1.2, 200)
        lo, hi = 40, int(4000.0 / (12000.0 / 2304)) - 1
        assert np.allclose(arr["sbase"][lo:hi], sbase[lo:hi], rtol=2e-3)              # the polynomial baseline (its own solver, single precision there)
        assert np.allclose(arr["savsm"][lo:hi], savsm[lo:hi], rtol=2e-3)
        gk = {c[0]: (c[3], c[2]) for c in got}
        wk = {c[0]: (c[1], c[2]) for c in want}
        total += len(wk)
        for key in set(gk) ^ set(wk):
            v = (gk.get(key) or wk.get(key))[1]
            flat = abs(savsm[key] - max(savsm[key - 1], savsm[key + 1])) <= 3e-3 * savsm[key]
            assert abs(v - 1.2) <= 5e-3 or flat, (k, key, v)
        for key in set(gk) & set(wk):
            assert gk[key][0] == pytest.approx(wk[key][0], abs=0.05) and gk[key][1] == pytest.approx(wk[key][1], rel=3e-3)
        # ORDER: the same open point as for FT8 -- the restatement lists the peaks strongest first, getcandidates4 as read here in ascending
        # frequency (it stops at maxcand while scanning upwards); the same entries unless the list is cut
        assert [c[2] for c in got] == sorted((c[2] for c in got), reverse=True)
    assert total > 3 * N_FRAMES


def _wspr_frame(seed, n_sig):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(1440000) * 200.0
    for _ in range(n_sig):
        a += wspr_audio(1440000, rng.uniform(1410, 1590), rng.uniform(0.5, 2.5), rng.uniform(60, 500), rng, drift_hz=rng.uniform(-2, 2))
    return to_i16(a)


def test_wspr_candidate_pick_agrees_with_the_independent_implementation(oracle):
    total = 0
    for k in range(N_FRAMES // 4):                                                     # (a 120 s frame takes the restatement ~1 s: 8 frames, 30+ peaks each)
        fr = _wspr_frame(3000 + k, n_sig=1 + k % 5)
        got, arr = oracle.wspr_search(fr, want_arrays=True)
        want, smspec = I.wspr_pick(arr["ps"])
        assert np.allclose(arr["smspec"], smspec, rtol=1e-4, atol=1e-5)
        # the restatement goes on to the coarse sync search, which moves a candidate by whole steps of df / 2 and keeps wsprd's order of the PICK:
        # compare the picks themselves -- every peak the independent pick finds appears, in the same (strongest-first) order, within the search's reach
        gf = [c[0] for c in got]
        total += len(want)
        assert len(got) == len(want), (k, len(got), len(want))
        for (f, snr), g in zip(want, got):
            assert abs(g[0] - f) <= 4 * 0.7325 and g[1] == pytest.approx(snr, abs=0.02), (k, f, snr, g)
    assert total >= 16


def _fst4w_frame(seed, n_sig):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(1440000) * 200.0
    for _ in range(n_sig):
        a += fst4w_audio(1440000, rng.uniform(1420, 1580), rng.uniform(0.3, 1.5), rng.uniform(30, 300), rng)
    return to_i16(a)


def test_fst4w_clean_loop_agrees_with_the_independent_implementation(oracle):
    total = 0
    df1, df2 = 12000.0 / 1440000, (12000.0 / 8200) / 2.0
    ina = int(np.floor(1400.0 / df2 + 0.5))
    first = int(np.floor(ina * df2 / df1 + 0.5)) - int(df2 / df1) // 2
    for k in range(N_FRAMES // 4):
        fr = _fst4w_frame(4000 + k, n_sig=1 + k % 6)
        got, arr = oracle.fst4w_candidates(fr, 1400, 1600, 1.2, want_arrays=True)
        want, s2 = I.fst4w_pick(arr["band"], first, 1400, 1600, 1.2)     # (the restatement hands back the band's power per bin)
        lo, hi = ina + 3, ina + 260
        assert np.allclose(arr["s2"][lo:hi], s2[lo:hi], rtol=2e-4)
        total += len(want)
        gb, wb = [c[2] for c in got], [c[0] for c in want]
        if gb != wb:                                                                   # the CLEAN loop is sequential: a tie at 1e-4 reorders two entries
            assert sorted(gb) == sorted(wb) or abs(len(gb) - len(wb)) <= 1, (k, gb, wb)
        for g, w in zip(got, want):
            if g[2] == w[0]:
                assert g[1] == pytest.approx(w[1], rel=5e-4)
    assert total >= 12
