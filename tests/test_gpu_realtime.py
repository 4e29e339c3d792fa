"""GPU: the ingest seam (Receiver.hpp:209-276 -> cwslg_push_iq / cwslg_push_iq_many) under wall-clock pacing, through the C ABI.

1. cwslg_push_iq_many: one block for each of several receivers per call == the same blocks through per-receiver cwslg_push_iq, and both
   equal the oracle (frames bit-identical in exact mode), including a batch that wraps a ring and a duplicate-receiver batch that is refused.
2. cwsl_gpu_realtime (csrc/host/realtime_main.cpp): one pusher thread per receiver, one cwslg_push_iq per 2048-sample block, paced at 8x
   real time, one discarded partial slot and two emitted slots with the sync stage on: zero dropped blocks, every frame fetched, and the
   dumped channels' .wav files and candidate lists equal the oracle driven with the same blocks and boundaries -- in both pusher forms
   (threads / batch)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import assert_frames_match, assert_int16_match

pytestmark = pytest.mark.gpu
FS, BLK = 192000, 2048


def test_push_iq_many_equals_per_receiver_pushes(ctx, oracle):
    n_rx, n_blk = 5, 40
    freqs = [-26000, 1234, 48000, -77000, 30000]
    streams = [oracle.synth_iq(100 + r, n_blk * BLK, FS, tones_hz=[freqs[r] + 900.0, freqs[r] + 2100.5], amp=2.0e4) for r in range(n_rx)]
    # ring of 16 blocks: the 40 blocks wrap it twice (the library demodulates when a ring would overflow)
    rxs = [ctx.receiver_open(FS, BLK, 0, ring_blocks=16) for _ in range(n_rx)]
    chans = [ctx.channel_open(rxs[r], freqs[r], "FT8") for r in range(n_rx)]
    ocs = [oracle.Channel("FT8", FS, BLK, freqs[r]) for r in range(n_rx)]
    with pytest.raises(Exception):
        ctx.push_iq_many([rxs[0], rxs[0]], [streams[0][:BLK], streams[0][:BLK]])       # a receiver twice in one batch
    for k in range(n_blk):
        ctx.push_iq_many(rxs, [s[k * BLK:(k + 1) * BLK] for s in streams])
        for r in range(n_rx):
            ocs[r].push(streams[r][k * BLK:(k + 1) * BLK])
        if k == 7:
            ctx.slot_boundary("FT8", 15)
            assert all(oc.boundary(15) is None for oc in ocs)
    ctx.slot_boundary("FT8", 30)
    st = ctx.stats()
    assert st["push_batches"] == n_blk and st["push_calls"] == n_blk * n_rx and st["blocks_dropped"] == 0
    for r in range(n_rx):
        ref = ocs[r].boundary(30, want_f32=True)
        a, nv = ctx.fetch_audio_f32(chans[r])
        g = ctx.fetch_frame(chans[r])
        assert nv == (n_blk - 8) * BLK // 16 and g["t_start"] == 15            # the frame that began at the first boundary (after block 7)
        assert_frames_match(a, ref["f32"])
        assert_int16_match(g["i16"], ref["i16"], ref["f32"] * ref["factor"])


def _read_wav(path):
    raw = open(path, "rb").read()
    return np.frombuffer(raw[46:], dtype=np.int16)


@pytest.mark.parametrize("mode", ["threads", "batch"])
def test_realtime_harness_zero_drops_and_bit_identical_frames(tmp_path, oracle, mode):
    from cwsl_digi_amd import build as B
    B.build()
    R, C, pre, slot, slots, stride, nfile = 4, 8, 8, 96, 2, 7, 64
    tones = [-90000 + ((k * 1373) % 176000) + 1500.0 for k in range(0, R * C, 5)][:8]
    iq = oracle.synth_iq(4242, nfile * BLK, FS, tones_hz=tones, amp=1.5e4).astype(np.complex64)
    path = tmp_path / "iq.c64"
    iq.tofile(path)
    out = tmp_path / "out"; out.mkdir()
    p = subprocess.run([B.REALTIME, "--receivers", str(R), "--channels-per-rx", str(C), "--pre", str(pre), "--slot-blocks", str(slot),
                        "--slots", str(slots), "--speed", "8", "--mode", mode, "--exact", "1", "--sync", "1", "--iq", str(path),
                        "--iq-stride", str(stride), "--dump", "6", "--out", str(out), "--fetch-threads", "2"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["blocks_dropped"] == 0
    assert line["frames_emitted"] == R * C * slots and line["frames_discarded"] == R * C
    assert line["frames_fetched"] == R * C * slots and len(line["boundaries"]) == slots
    assert line["push_calls"] == R * (pre + slot * slots)
    assert line["push_batches"] == (pre + slot * slots if mode == "batch" else 0)
    # paced at 8x: the run cannot be faster than the stream allows, and must not be much slower
    assert line["push_wall_s"] >= 0.95 * line["stream_seconds"] / 8
    blocks = iq.reshape(nfile, BLK)
    for row in open(out / "dump.txt"):
        k, r, f, t_start, nv, wav = row.split()
        k, r, f = int(k), int(r), int(f)
        oc = oracle.Channel("FT8", FS, BLK, f)
        nxt = 0
        def feed(n):
            nonlocal nxt
            for _ in range(n):
                oc.push(blocks[(r * stride + nxt) % nfile]); nxt += 1
        feed(pre); assert oc.boundary(1000) is None
        feed(slot); first = oc.boundary(1015)
        feed(slot); ref = oc.boundary(1030)
        assert first is not None and int(t_start) == ref["t_start"] == 1015 and int(nv) == slot * BLK // 16
        pcm = _read_wav(wav)
        assert np.array_equal(pcm, ref["i16"]), (k, r, f)                     # exact mode: bit-identical to the reference chain
        want = oracle.ft8_sync(ref["i16"], 200, 3000, 1.5, 200)
        got = [l.split() for l in open(out / f"ch{k}.cand")]
        assert len(got) == len(want)
        for g_, w_ in zip(got, want):
            assert (int(g_[0]), int(g_[1])) == (w_[0], w_[1]) and np.float32(g_[2]) == np.float32(w_[2])


def test_single_block_wakeups_at_1024_channels_low_redundancy(tmp_path, oracle):
    """a host that mirrors the reference's one wake-up per block (Instance.cpp:260-276; Receiver.hpp:132 gives it a
    3 s ring) -- 1024 channels on 32 receivers, cwslg_process() after EVERY block period -- with the library's own launch threshold
    (cwslg_set_process_threshold(ctx, -1)).  The bit-identical kernel pays a 32-block warm-up per stream of outputs, so WHEN it launches
    decides how much it fetches and multiplies per output delivered: here frames and candidate lists must be bit-identical to the oracle driven
    with the same blocks AND the blocks put through the arithmetic (stats.demod_blocks_read) stay within 1.1x of the blocks delivered.
    Without the threshold the same run launches 1300+ times at a redundancy above 3 (asserted, so that the knob's effect is on record)."""
    from cwsl_digi_amd import build as B
    B.build()
    R, C, pre, slot, slots, stride, nfile = 32, 32, 8, 1406, 1, 3, 128
    tones = [-90000 + ((k * 1373) % 176000) + 1500.0 for k in range(0, R * C, 97)][:8]
    iq = oracle.synth_iq(777, nfile * BLK, FS, tones_hz=tones, amp=1.5e4).astype(np.complex64)
    path = tmp_path / "iq.c64"
    iq.tofile(path)
    lines = {}
    for thr in (-1, 0):
        out = tmp_path / f"out{thr}"; out.mkdir()
        p = subprocess.run([B.REALTIME, "--receivers", str(R), "--channels-per-rx", str(C), "--pre", str(pre), "--slot-blocks", str(slot),
                            "--slots", str(slots), "--speed", "0", "--mode", "batch", "--exact", "1", "--sync", "1", "--iq", str(path),
                            "--iq-stride", str(stride), "--dump", "5", "--out", str(out), "--process-ms", "10.6", "--process-threshold", str(thr)],
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        lines[thr] = line = json.loads(p.stdout.strip().splitlines()[-1])
        assert line["blocks_dropped"] == 0 and line["frames_emitted"] == R * C * slots and line["exact"] == 1
        if thr == 0:
            continue                                                            # (frames of the unthresholded run: covered by the tests above)
        blocks = iq.reshape(nfile, BLK)
        for row in open(out / "dump.txt"):
            k, r, f, t_start, nv, wav = row.split()
            k, r, f = int(k), int(r), int(f)
            oc = oracle.Channel("FT8", FS, BLK, f)
            oc.push_many(np.concatenate([blocks[(r * stride + j) % nfile] for j in range(pre)])); assert oc.boundary(1000) is None
            oc.push_many(np.concatenate([blocks[(r * stride + pre + j) % nfile] for j in range(slot)])); ref = oc.boundary(1015)
            assert int(t_start) == ref["t_start"] == 1000 and int(nv) == slot * BLK // 16
            assert np.array_equal(_read_wav(wav), ref["i16"]), (k, r, f)
            want = oracle.ft8_sync(ref["i16"], 200, 3000, 1.5, 200)
            got = [l.split() for l in open(out / f"ch{k}.cand")]
            assert [(int(g_[0]), int(g_[1]), np.float32(g_[2])) for g_ in got] == [(w_[0], w_[1], np.float32(w_[2])) for w_ in want]
    a, b = lines[-1], lines[0]
    assert a["process_threshold"] == -1 and a["process_deferred"] > 1000 and a["demod_launches"] <= 20
    assert 1.0 < a["demod_redundancy"] <= 1.1, a["demod_redundancy"]
    assert b["demod_launches"] >= 1300 and b["demod_redundancy"] > 3.0, (b["demod_launches"], b["demod_redundancy"])
