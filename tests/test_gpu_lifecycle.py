"""GPU: runtime open/close (band rotation, CWSL_DIGI.cpp:1217-1226), concurrent callers, long-running slots."""
import threading
import time

import numpy as np
import pytest

import cwsl_digi_amd as P
from conftest import assert_frames_match

pytestmark = pytest.mark.gpu
FS, BLK = 192000, 2048


def test_reopen_channels_and_receivers_many_times(ctx, oracle):
    """Channels and receivers come and go at runtime; ids are recycled; results stay right."""
    iq = oracle.synth_iq(1, 40 * BLK, FS, tones_hz=[5000 + 1000.0], amp=1e4)
    ref = oracle.Demod(FS, 5000).run(iq)
    for rep in range(12):
        rx = ctx.receiver_open(FS, BLK, 0)
        chans = [ctx.channel_open(rx, 5000, m) for m in ("FT8", "FT4", "WSPR")]
        for ch, g in zip(chans, ("FT8", "FT4", "S120")):
            ctx.slot_boundary(g, 1)
        ctx.push_iq(rx, iq)
        for g in ("FT8", "FT4", "S120"):
            ctx.slot_boundary(g, 2)
        for ch in chans:
            a, nv = ctx.fetch_audio_f32(ch)
            assert nv == len(ref)
            assert_frames_match(a[:nv], ref)
        if rep % 2:
            for ch in chans:
                ctx.channel_close(ch)
        ctx.receiver_close(rx)                    # closes whatever channels are still open (Receiver::finish)
    assert ctx.stats()["frames_emitted"] == 36


def test_concurrent_push_boundary_fetch(ctx, oracle):
    """{Receiver thread} || {slot-clock thread} || {consumer}: every entry point is safe under one mutex."""
    n_rx, blocks = 4, 200
    rxs = [ctx.receiver_open(FS, BLK, 0) for _ in range(n_rx)]
    chans = [ctx.channel_open(rx, -20000 + 9000 * k, "FT8") for k, rx in enumerate(rxs)]
    blk = [oracle.synth_iq(100 + k, BLK, FS) for k in range(n_rx)]
    errors = []
    stop = threading.Event()

    def pusher(k):
        try:
            for _ in range(blocks):
                ctx.push_iq(rxs[k], blk[k])
        except Exception as e:       # pragma: no cover
            errors.append(e)

    def clock():
        t = 1
        try:
            while not stop.is_set():
                ctx.slot_boundary("FT8", t); t += 1
        except Exception as e:       # pragma: no cover
            errors.append(e)

    def consumer():
        try:
            while not stop.is_set():
                for ch in chans:
                    fr = ctx.fetch_frame(ch)
                    assert fr is None or len(fr["i16"]) == 240000
        except Exception as e:       # pragma: no cover
            errors.append(e)

    th = [threading.Thread(target=pusher, args=(k,)) for k in range(n_rx)] + [threading.Thread(target=clock), threading.Thread(target=consumer)]
    for t in th:
        t.start()
    for t in th[:n_rx]:
        t.join()
    stop.set()
    for t in th[n_rx:]:
        t.join()
    assert not errors, errors
    ctx.slot_boundary("FT8", 10 ** 6)
    st = ctx.stats()
    assert st["demod_samples"] == n_rx * blocks * BLK      # every pushed sample was demodulated exactly once


def test_repeated_discards_keep_the_phasor_running(ctx, oracle):
    """Boundaries stamped 0 discard their frame WITHOUT restarting the demodulator (Instance.cpp:224-227 `continue`
    skips :251), so the phasor recurrence and the filter history run on from channel creation for as many slots as
    that lasts.  Three discarded 7.5 s slots put the block count past the checkpoint table sized at open (two frames):
    the table is extended (stats: phasor_regrows) and the first emitted frame still equals the oracle driven with the
    same pushes and boundaries."""
    n = 1440000 // BLK * BLK                       # one FT4 slot of whole blocks
    f = 31000
    iq = oracle.synth_iq(77, 4 * n, FS, tones_hz=[f + 900.0, f + 2222.0], amp=1.2e4)
    rx = ctx.receiver_open(FS, BLK, 0)
    ch = ctx.channel_open(rx, f, "FT4")
    oc = oracle.Channel("FT4", FS, BLK, f)
    epochs = [0, 0, 0, 7, 15]                      # frames started at epoch 0 are discarded when they end
    ctx.slot_boundary("FT4", epochs[0]); assert oc.boundary(epochs[0]) is None
    for part in range(4):
        seg = iq[part * n:(part + 1) * n]
        for k in range(0, n, 100 * BLK):
            ctx.push_iq(rx, seg[k:k + 100 * BLK])
        oc.push_many(seg)
        ctx.slot_boundary("FT4", epochs[part + 1])
        r = oc.boundary(epochs[part + 1], want_f32=True)
        g = ctx.fetch_frame(ch)
        if part < 3:
            assert r is None and g is None         # discarded on both sides
        else:
            a, nv = ctx.fetch_audio_f32(ch)
            assert g["t_start"] == r["t_start"] == 7 and nv == n // 16
            assert_frames_match(a, r["f32"])
    st = ctx.stats()
    assert st["frames_discarded"] == 4 and st["frames_emitted"] == 1 and st["phasor_regrows"] >= 1


def test_late_first_boundary_saturates_then_recovers(ctx, oracle):
    """No boundary for longer than period + 5 s after open: the frame fills up, later blocks are dropped ("af buffer
    full", Instance.cpp:268-271), the late boundary discards the partial-slot frame, and the next slot is demodulated
    from a valid checkpoint range (no read past the phasor table) -- its frame is finite and spectrally right."""
    n = 2000 * BLK                                  # 21.3 s > FT4's 12.5 s frame
    f = -12000
    rx = ctx.receiver_open(FS, BLK, 0, ring_blocks=1300)
    ch = ctx.channel_open(rx, f, "FT4")
    for k in range(0, n, 250 * BLK):
        ctx.push_synth(rx, 5, 250 * BLK, BLK, tones_hz=[f + 1000.0], amp=1e4)
    assert ctx.stats()["blocks_dropped"] > 0
    ctx.slot_boundary("FT4", 0)                     # late first boundary: discard
    assert ctx.fetch_frame(ch) is None
    ctx.push_synth(rx, 5, 600 * BLK, BLK, tones_hz=[f + 1000.0], amp=1e4)
    ctx.slot_boundary("FT4", 8)                     # frame started at epoch 0: discarded as well, demodulator keeps running
    ctx.push_synth(rx, 5, 600 * BLK, BLK, tones_hz=[f + 1000.0], amp=1e4)
    ctx.slot_boundary("FT4", 16)
    a, nv = ctx.fetch_audio_f32(ch)
    assert nv == 600 * BLK // 16 and np.isfinite(a).all()
    spec = np.abs(np.fft.rfft(a[2000:2000 + 48000].astype(np.float64)))
    assert abs(int(np.argmax(spec)) * 12000.0 / 48000 - 1000.0) < 1.0      # the 1 kHz tone, at the right audio pitch


@pytest.mark.parametrize("mode", ["FST4W-1800", "FST4-900"])
def test_longest_modes_allocate_and_frame(ctx, oracle, mode):
    """The 900 s / 1800 s modes: frames of 12000*(period+5) samples (1800 s: 21.66 M int16 = 43 MB, two float frames of 87 MB and
    87 MB of phasor checkpoints per channel).  A short slot is enough to exercise the allocation, the framing and the zero tail."""
    import cwsl_digi_amd as P
    n = 96 * BLK                                   # 1 s of IQ
    f = 1500
    iq = oracle.synth_iq(9, 2 * n, FS, tones_hz=[f + 1500.0], amp=1e4)
    rx = ctx.receiver_open(FS, BLK, 0)
    ch = ctx.channel_open(rx, f, mode)
    oc = oracle.Channel(mode, FS, BLK, f)
    g = P.group_of(mode)
    ctx.push_iq(rx, iq[:n]); oc.push_many(iq[:n])
    ctx.slot_boundary(g, 1800); assert oc.boundary(1800) is None and ctx.fetch_frame(ch) is None
    ctx.push_iq(rx, iq[n:]); oc.push_many(iq[n:])
    ctx.slot_boundary(g, 3600)
    r = oc.boundary(3600, want_f32=True)
    fr = ctx.fetch_frame(ch)
    a, nv = ctx.fetch_audio_f32(ch)
    assert len(fr["i16"]) == P.frame_len(mode) == 12000 * (int(mode.split("-")[1]) + 5) and nv == n // 16 and fr["t_start"] == 1800
    assert_frames_match(a[:nv + 64], r["f32"][:nv + 64])
    assert not fr["i16"][nv:].any()                # the reference's zero tail
    d = np.abs(fr["i16"][:nv].astype(np.int32) - r["i16"][:nv].astype(np.int32))
    assert d.max() <= 1


def test_fetch_is_atomic_against_the_next_boundary(ctx, oracle):
    """A frame is handed out whole: the reference copies it into the ItemToDecode it pushes (Instance.cpp:238-245, DecoderPool.hpp:174-210),
    so a consumer that is late never sees samples of slot N + 1 under the start epoch of slot N.  Here a clock thread pushes a short slot
    of distinct content and fires the boundary, back to back, on 64 channels, while three consumer threads fetch in a loop; every frame
    that comes back must be -- bit for bit (the default mode is the bit-identical one) -- the oracle's frame of the start_epoch it was
    returned with.  (Round 4's library queued the D2H copy with the context mutex released and nothing kept the next boundary's
    finalize_kernel from rewriting the buffer under it.)"""
    n_ch, n_epochs, n_blk = 64, 24, 6
    n_live = n_blk * BLK // 16 + 64                        # samples past this index are the frame's zero tail
    rx = ctx.receiver_open(FS, BLK, 0)
    freqs = [-88000 + 2750 * k for k in range(n_ch)]
    chans = [ctx.channel_open(rx, f, "FT8") for f in freqs]
    segs = {e: oracle.synth_iq(5000 + e, n_blk * BLK, FS, tones_hz=[freqs[(7 * e) % n_ch] + 700.0 + 13.0 * e, freqs[(3 * e + 1) % n_ch] + 1500.0], amp=1.5e4)
            for e in range(1, n_epochs + 1)}
    want = {}                                              # (channel index, start epoch) -> the live part of the int16 frame
    for k, f in enumerate(freqs):
        oc = oracle.Channel("FT8", FS, BLK, f)
        assert oc.boundary(1) is None                      # the first, partial slot is discarded (Instance.cpp:224-227)
        for e in range(1, n_epochs + 1):
            oc.push_many(segs[e])
            r = oc.boundary(e + 1)
            assert r["t_start"] == e
            assert not r["i16"][n_live:].any()
            want[(k, e)] = r["i16"][:n_live].astype(np.int32)
        oc.close()
    for k in range(n_ch):                                  # the frames really differ from epoch to epoch, by far more than the fast mode's +-1 LSB
        for e in range(1, n_epochs):
            assert np.abs(want[(k, e)] - want[(k, e + 1)]).max() > 100
    lsb = 0 if ctx.mode == "exact" else 1                  # exact (default) mode: the reference's bits; fast mode: rounding ties may differ
    errors, seen = [], set()
    stop = threading.Event()

    def clock():
        try:
            ctx.slot_boundary("FT8", 1)
            for e in range(1, n_epochs + 1):
                ctx.push_iq(rx, segs[e])
                ctx.slot_boundary("FT8", e + 1)
        except Exception as ex:       # pragma: no cover
            errors.append(ex)
        finally:
            stop.set()

    def consumer(first):
        try:
            k = first
            while True:
                done = stop.is_set()                       # one more round after the clock has finished
                for _ in range(n_ch):
                    fr = ctx.fetch_frame(chans[k])
                    if fr is not None:
                        e = int(fr["t_start"])
                        if np.abs(fr["i16"][:n_live].astype(np.int32) - want[(k, e)]).max() > lsb or fr["i16"][n_live:].any():
                            errors.append(AssertionError(f"channel {k}: frame returned with start epoch {e} is not that slot's frame"))
                            return
                        seen.add((k, e))
                    k = (k + 1) % n_ch
                if done:
                    return
        except Exception as ex:       # pragma: no cover
            errors.append(ex)

    th = [threading.Thread(target=clock)] + [threading.Thread(target=consumer, args=(17 * j,)) for j in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]
    assert {e for _, e in seen} >= {n_epochs} and len({e for _, e in seen}) >= 3     # the consumers overlapped several generations


def test_candidate_lists_are_atomic_against_the_next_boundary(ctx, oracle):
    """ABI 5: the results of one epoch, atomically.  ItemToDecode carries epochTime with its audio (DecoderPool.hpp:174-210); here the candidate
    lists travel the same way.  A clock thread pushes whole 48 kHz FT8 slots of distinct noise + tones and fires the boundary, back to back, while
    three consumers loop over (a) cwslg_fetch_slot -- the frame must be the oracle's frame of the start epoch it is returned with AND the list
    must be the restated search of exactly that frame -- and (b) cwslg_fetch_candidates with its start_epoch: in exact mode the list must be the
    oracle's list of the oracle's frame of that epoch.  (ABI 4 returned lists without an epoch and copied them under the context mutex on the
    compute stream; a consumer one boundary late paired frame N with list N + 1 and could not tell.)"""
    fs, blk = 48000, 512
    n_ch, n_epochs, n_blk = 6, 9, 1400                     # 1400 x 512 / 48000 = 14.9 s of signal per slot
    ctx.enable_sync(True, 1.5, 100, 200, 3000)
    rx = ctx.receiver_open(fs, blk, 0)
    freqs = [-18000 + 6000 * k for k in range(n_ch)]
    chans = [ctx.channel_open(rx, f, "FT8") for f in freqs]
    segs = {e: oracle.synth_iq(7000 + e, n_blk * blk, fs, tones_hz=[freqs[e % n_ch] + 700.0 + 13.0 * e, freqs[(3 * e + 1) % n_ch] + 1500.0], amp=1.5e4)
            for e in range(1, n_epochs + 1)}
    want, want_list = {}, {}
    for k, f in enumerate(freqs):
        oc = oracle.Channel("FT8", fs, blk, f)
        assert oc.boundary(1) is None
        for e in range(1, n_epochs + 1):
            oc.push_many(segs[e])
            r = oc.boundary(e + 1)
            want[(k, e)] = r["i16"].copy()
            want_list[(k, e)] = oracle.ft8_sync(r["i16"], 200, 3000, 1.5, 100)
        oc.close()
    key = lambda lst: [(c[0], c[1], np.float32(c[2]).view(np.uint32)) for c in lst]
    for k in range(n_ch):                                  # the lists really differ from epoch to epoch
        assert len({tuple(key(want_list[(k, e)])) for e in range(1, n_epochs + 1)}) >= n_epochs - 1
    exact = ctx.mode == "exact"
    errors, seen = [], set()
    stop = threading.Event()

    def clock():
        try:
            ctx.slot_boundary("FT8", 1)
            for e in range(1, n_epochs + 1):
                ctx.push_iq(rx, segs[e])
                ctx.slot_boundary("FT8", e + 1)
                time.sleep(0.35)                            # (a consumer's round costs six runs of the restatement: let it meet several generations)
        except Exception as ex:       # pragma: no cover
            errors.append(ex)
        finally:
            stop.set()

    def consumer(first):
        try:
            k = first
            while True:
                done = stop.is_set()
                for _ in range(n_ch):
                    r = ctx.fetch_slot(chans[k], max_list=100)
                    if r is not None:
                        e = int(r["t_start"])
                        if np.abs(r["i16"].astype(np.int32) - want[(k, e)].astype(np.int32)).max() > (0 if exact else 1):
                            errors.append(AssertionError(f"fetch_slot, channel {k}: frame returned with start epoch {e} is not that slot's frame")); return
                        if r["list_kind"] != "FT8" or key(r["list"]) != key(oracle.ft8_sync(r["i16"], 200, 3000, 1.5, 100)):
                            errors.append(AssertionError(f"fetch_slot, channel {k}, epoch {e}: the list is not the search of the frame it came with")); return
                        seen.add((k, e))
                    try:
                        lst, e = ctx.fetch_candidates(chans[k], 100, with_epoch=True)
                    except P.CwslGpuError as ex:
                        if ex.status != -9:                    # CWSLG_ERR_NO_FRAME
                            raise
                        lst = None                          # no list yet (first, discarded slot)
                    if lst is not None and exact and key(lst) != key(want_list[(k, int(e))]):
                        errors.append(AssertionError(f"fetch_candidates, channel {k}: list returned with start epoch {e} is not that slot's list")); return
                    k = (k + 1) % n_ch
                if done:
                    return
        except Exception as ex:       # pragma: no cover
            errors.append(ex)

    th = [threading.Thread(target=clock)] + [threading.Thread(target=consumer, args=(2 * j,)) for j in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]
    assert {e for _, e in seen} >= {n_epochs} and len({e for _, e in seen}) >= 3


def test_process_threshold_defers_and_flush_overrides(ctx, oracle):
    """cwslg_set_process_threshold / cwslg_flush (ABI 5): below the threshold a cwslg_process() over a few pending blocks returns without a
    launch; cwslg_flush demodulates them at once; the boundary takes whatever is left.  The frame is the oracle's whichever way the launches fell."""
    f, n_blk = 12000, 96
    iq = oracle.synth_iq(77, n_blk * BLK, FS, tones_hz=[f + 1100.0, f + 1900.5], amp=1.2e4)
    rx = ctx.receiver_open(FS, BLK, 0)
    ch = ctx.channel_open(rx, f, "FT8")
    oc = oracle.Channel("FT8", FS, BLK, f)
    ctx.slot_boundary("FT8", 15); assert oc.boundary(15) is None
    ctx.set_process_threshold(20480 if ctx.mode == "exact" else 100000)      # (exact mode: the value the library itself picks with -1)
    base = ctx.stats()
    for k in range(0, 40):
        ctx.push_iq(rx, iq[k * BLK:(k + 1) * BLK]); ctx.process()
    st = ctx.stats()
    assert st["process_deferred"] - base["process_deferred"] == 40 and st["demod_launches"] == base["demod_launches"]
    ctx.flush()
    assert ctx.stats()["demod_launches"] == base["demod_launches"] + 1
    for k in range(40, n_blk):
        ctx.push_iq(rx, iq[k * BLK:(k + 1) * BLK]); ctx.process()
    ctx.slot_boundary("FT8", 30)
    st = ctx.stats()
    assert st["demod_launches"] == base["demod_launches"] + 2 and st["demod_samples"] - base["demod_samples"] == n_blk * BLK
    assert 1.0 < (st["demod_blocks_read"] - base["demod_blocks_read"]) * 16 / (n_blk * BLK) < 12.0
    oc.push_many(iq); ref = oc.boundary(30, want_f32=True)
    a, nv = ctx.fetch_audio_f32(ch)
    assert nv == n_blk * BLK // 16
    assert_frames_match(a[:nv], ref["f32"][:nv])
    ctx.set_process_threshold(0)
