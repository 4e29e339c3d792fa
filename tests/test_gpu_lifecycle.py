"""GPU: runtime open/close (band rotation, CWSL_DIGI.cpp:1217-1226), concurrent callers, long-running slots."""
import threading

import numpy as np
import pytest

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
