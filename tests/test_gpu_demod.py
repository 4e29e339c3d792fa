"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded input.

Tolerance (BASELINE.json north_star): float audio within 1e-5 of frame peak; int16 frames identical
except documented +-1 LSB rounding ties; phasor checkpoints and host constants bit-exact.
"""
import numpy as np
import pytest

from conftest import assert_frames_match, assert_int16_match

pytestmark = pytest.mark.gpu


FS = 192000
IQ_LEN = 2048
FREQS = [0, 1234, 24000, 87000, -50000, -93000, -26000]


def _tones(f):
    return [f + 700.0, f + 1500.5, f + 2600.25]


@pytest.mark.parametrize("fs", [192000, 96000, 48000])
def test_host_constants_bit_exact(ctx, oracle, fs):
    """taps / tone / phase_inc uploaded to the GPU are the reference's bits (SSBD.hpp:62-68,110-114)."""
    rx = ctx.receiver_open(fs, 1024, 0)
    for f in [0, 1234, -9000, 17000]:
        ch = ctx.channel_open(rx, f, "FT8")
        taps, tone, inc = ctx.channel_constants(ch)
        d = oracle.Demod(fs, float(np.float32(f)))
        assert np.array_equal(taps.view(np.uint32), d.taps.view(np.uint32))
        assert np.array_equal(tone.view(np.uint32), d.tone.view(np.uint32))
        assert np.array_equal(np.array([inc]).view(np.uint32), np.array([d.phase_inc]).view(np.uint32))
        ctx.channel_close(ch)


@pytest.mark.parametrize("f", FREQS)
def test_phasor_checkpoints_bit_exact(ctx, oracle, f):
    """Device-built checkpoints == the float32 recurrence phase *= phase_inc (SSBD.hpp:174)."""
    rx = ctx.receiver_open(FS, IQ_LEN, 0)
    ch = ctx.channel_open(rx, f, "FT8")
    st = ctx.checkpoint_stride()
    nblk = 192000                                      # 16 s
    ck = ctx.phasor_checkpoints(ch, nblk // st)
    d = oracle.Demod(FS, f)
    _, tr = d.run(np.zeros(nblk * 16, np.complex64), trace=True)
    assert len(ck) == nblk // st
    assert np.array_equal(ck.view(np.uint64), tr[::st][:nblk // st].view(np.uint64))


def _run_gpu_slot(ctx, rx, chans, iq_a, iq_b, block=IQ_LEN):
    """partial slot iq_a, boundary (discard), full slot iq_b, boundary (emit)."""
    for k in range(0, len(iq_a), block):
        ctx.push_iq(rx, iq_a[k:k + block])
    ctx.slot_boundary("FT8", 1000)
    for c in chans:
        assert ctx.fetch_frame(c) is None
    # push the second slot in uneven batches to exercise pending accumulation + ring wrap
    k = 0
    step = [block, 3 * block, 7 * block, block]
    j = 0
    while k < len(iq_b):
        n = min(step[j % 4], len(iq_b) - k)
        ctx.push_iq(rx, iq_b[k:k + n])
        if j % 3 == 0:
            ctx.process()
        k += n
        j += 1
    ctx.slot_boundary("FT8", 1015)


def _run_oracle_slot(oracle, f, iq_a, iq_b, mode="FT8", fs=FS, block=IQ_LEN):
    c = oracle.Channel(mode, fs, block, f)
    c.push_many(iq_a)
    assert c.boundary(1000) is None
    c.push_many(iq_b)
    r = c.boundary(1015, want_f32=True)
    assert r is not None
    return r


def test_demod_parity_multi_channel_shared_receiver(ctx, oracle):
    """7 channels on ONE receiver (the reference topology), first emitted frame continues the
    demodulator across the discarded partial slot (Instance.cpp:224-227 has no SSBD reset)."""
    na, nb = 40 * IQ_LEN, 150 * IQ_LEN
    tones = sum((_tones(f) for f in FREQS), [])
    iq = oracle.synth_iq(0xC0FFEE, na + nb, FS, tones_hz=tones, amp=2.0e4)
    rx = ctx.receiver_open(FS, IQ_LEN, 0)
    chans = [ctx.channel_open(rx, f, "FT8") for f in FREQS]
    _run_gpu_slot(ctx, rx, chans, iq[:na], iq[na:])
    for f, ch in zip(FREQS, chans):
        ref = _run_oracle_slot(oracle, f, iq[:na], iq[na:])
        got = ctx.fetch_frame(ch)
        f32, nv = ctx.fetch_audio_f32(ch)
        assert got["t_start"] == 1000 and ref["t_start"] == 1000
        assert nv == nb // 16 == got["n_valid"]
        rel = assert_frames_match(f32, ref["f32"])
        assert rel < 2e-6, rel                           # expected ~4e-7
        assert abs(float(got["factor"]) - float(ref["factor"])) <= 2e-6 * float(ref["factor"])
        scaled = ref["f32"] * ref["factor"]
        assert_int16_match(got["i16"], ref["i16"], scaled)
        assert not got["i16"][nv:].any()                 # the reference's zero tail


def test_demod_parity_second_frame_fresh_demodulator(ctx, oracle):
    """Second emitted frame: new SSBD at the boundary -> zero history, phasor (1,0) (Instance.cpp:251)."""
    n1, n2, n3 = 8 * IQ_LEN, 64 * IQ_LEN, 96 * IQ_LEN
    f = -26000
    iq = oracle.synth_iq(77, n1 + n2 + n3, FS, tones_hz=_tones(f), amp=2.0e4)
    rx = ctx.receiver_open(FS, IQ_LEN, 0)
    ch = ctx.channel_open(rx, f, "FT8")
    oc = oracle.Channel("FT8", FS, IQ_LEN, f)
    ctx.push_iq(rx, iq[:n1]); oc.push_many(iq[:n1])
    ctx.slot_boundary("FT8", 15); assert oc.boundary(15) is None
    ctx.push_iq(rx, iq[n1:n1 + n2]); oc.push_many(iq[n1:n1 + n2])
    ctx.slot_boundary("FT8", 30); r1 = oc.boundary(30, want_f32=True)
    g1 = ctx.fetch_frame(ch); a1, _ = ctx.fetch_audio_f32(ch)
    ctx.push_iq(rx, iq[n1 + n2:]); oc.push_many(iq[n1 + n2:])
    ctx.slot_boundary("FT8", 45); r2 = oc.boundary(45, want_f32=True)
    g2 = ctx.fetch_frame(ch); a2, nv2 = ctx.fetch_audio_f32(ch)
    assert g1["t_start"] == 15 and g2["t_start"] == 30 and r2["t_start"] == 30
    assert_frames_match(a1, r1["f32"])
    assert_frames_match(a2, r2["f32"])
    assert nv2 == n3 // 16
    assert_int16_match(g2["i16"], r2["i16"], r2["f32"] * r2["factor"])


@pytest.mark.parametrize("fs,block", [(96000, 1024), (48000, 512)])
def test_demod_parity_other_rates(ctx, oracle, fs, block):
    f = 5000
    na, nb = 16 * block, 200 * block
    iq = oracle.synth_iq(5, na + nb, fs, tones_hz=[f + 900.0, f + 2100.0], amp=1.5e4)
    rx = ctx.receiver_open(fs, block, 0)
    ch = ctx.channel_open(rx, f, "FT4")
    for k in range(0, na, block):
        ctx.push_iq(rx, iq[k:k + block])
    ctx.slot_boundary("FT4", 7)
    ctx.push_iq(rx, iq[na:])
    ctx.slot_boundary("FT4", 14)
    oc = oracle.Channel("FT4", fs, block, f)
    oc.push_many(iq[:na]); assert oc.boundary(7) is None
    oc.push_many(iq[na:]); r = oc.boundary(14, want_f32=True)
    a, nv = ctx.fetch_audio_f32(ch)
    g = ctx.fetch_frame(ch)
    assert nv == nb // (fs // 12000)
    assert_frames_match(a, r["f32"])
    assert_int16_match(g["i16"], r["i16"], r["f32"] * r["factor"])


def test_lsb_channel(ctx, oracle):
    """LSB tuning (SSBD.hpp:110-111,133,135): sign flips the Im outputs and the B/2 offset."""
    f = 12000
    n = 80 * IQ_LEN
    iq = oracle.synth_iq(21, n, FS, tones_hz=[f - 800.0, f - 2200.0], amp=1.0e4)
    rx = ctx.receiver_open(FS, IQ_LEN, 0)
    ch = ctx.channel_open(rx, f, "FT8", usb=False)
    ctx.slot_boundary("FT8", 15)
    ctx.push_iq(rx, iq)
    ctx.slot_boundary("FT8", 30)
    a, nv = ctx.fetch_audio_f32(ch)
    ref = oracle.Demod(FS, f, usb=False).run(iq)
    assert nv == len(ref)
    assert_frames_match(a[:nv], ref)


def test_wspr_scale_rule(ctx, oracle):
    """WSPR uses wspraudioscalefactor, FST4W-120 the FT factor (exact "WSPR" compare, Instance.cpp:320)."""
    f = 1500
    n = 64 * IQ_LEN
    iq = oracle.synth_iq(9, 2 * n, FS, tones_hz=[f + 1500.0], amp=1.0e4)
    rx = ctx.receiver_open(FS, IQ_LEN, 0)
    chw = ctx.channel_open(rx, f, "WSPR")
    chf = ctx.channel_open(rx, f, "FST4W-120")
    ctx.push_iq(rx, iq[:n]); ctx.slot_boundary("S120", 120)
    ctx.push_iq(rx, iq[n:]); ctx.slot_boundary("S120", 240)
    for ch, mode in ((chw, "WSPR"), (chf, "FST4W-120")):
        oc = oracle.Channel(mode, FS, IQ_LEN, f)
        oc.push_many(iq[:n]); oc.boundary(120)
        oc.push_many(iq[n:]); r = oc.boundary(240, want_f32=True)
        g = ctx.fetch_frame(ch)
        assert len(g["i16"]) == 1500000
        assert abs(float(g["factor"]) - float(r["factor"])) <= 2e-6 * float(r["factor"])
        assert_int16_match(g["i16"], r["i16"], r["f32"] * r["factor"])
    assert float(ctx.fetch_frame(chw)["factor"]) < 0.3 * float(ctx.fetch_frame(chf)["factor"])


def test_frame_overflow_guard(ctx, oracle):
    """Blocks that would overflow the frame are dropped exactly like Instance.cpp:268-271
    (audio fill + IQ block length compared against size-1)."""
    f = 3000
    blk = 2048
    n_slot = 118 * 2048 * 16                 # 241664 outputs worth of input: more than the 240000 frame
    iq = oracle.synth_iq(3, 4 * blk + n_slot, FS, tones_hz=[f + 1000.0], amp=1e4)
    rx = ctx.receiver_open(FS, blk, 0)
    ch = ctx.channel_open(rx, f, "FT8")
    oc = oracle.Channel("FT8", FS, blk, f)
    ctx.push_iq(rx, iq[:4 * blk]); oc.push_many(iq[:4 * blk])
    ctx.slot_boundary("FT8", 15); oc.boundary(15)
    for k in range(4 * blk, len(iq), 64 * blk):
        ctx.push_iq(rx, iq[k:k + 64 * blk]); oc.push_many(iq[k:k + 64 * blk])
    fill = oc.fill
    ctx.slot_boundary("FT8", 30); r = oc.boundary(30, want_f32=True)
    a, nv = ctx.fetch_audio_f32(ch)
    assert oc.dropped > 0 and ctx.stats()["blocks_dropped"] == oc.dropped
    assert nv == fill and fill + blk > 240000 - 1
    assert_frames_match(a, r["f32"])


def test_errors_mirror_reference(ctx):
    import cwsl_digi_amd as P
    rx = ctx.receiver_open(FS, IQ_LEN, 0)
    with pytest.raises(P.CwslGpuError) as e:
        ctx.channel_open(rx, 97000, "FT8")          # |F| > Fs/2
    assert e.value.status == -2
    with pytest.raises(P.CwslGpuError) as e:
        ctx.channel_open(rx, 93000, "FT8")          # |F+B| > Fs/2
    assert e.value.status == -3
    with pytest.raises(P.CwslGpuError) as e:
        ctx.channel_open(rx, 0, "PSK31")
    assert e.value.status == -5
    with pytest.raises(P.CwslGpuError) as e:
        ctx.receiver_open(44100, 1024, 0)
    assert e.value.status in (-1, -10)
    with pytest.raises(P.CwslGpuError) as e:
        ctx.push_iq(rx, np.zeros(100, np.complex64))
    assert e.value.status == -11


def test_synth_source_matches_oracle_and_private_streams(ctx, oracle):
    """Device-side synthetic source is bit-identical to the oracle's generator, so full-size runs can be
    re-derived on the host.  16 private-stream slots (one receiver each), whole-slot batch."""
    n = 96 * IQ_LEN
    chans, rxs = [], []
    for s in range(16):
        f = -90000 + 11000 * s
        rx = ctx.receiver_open(FS, IQ_LEN, 0, ring_blocks=n // IQ_LEN + 8)
        ch = ctx.channel_open(rx, f, "FT8")
        ctx.slot_boundary_channel(ch, 15)        # discard the (empty) partial slot
        ctx.push_synth(rx, 0xC0FFEE ^ s, n, IQ_LEN, tones_hz=_tones(f), amp=2.0e4)
        rxs.append(rx); chans.append((ch, f, s))
    ctx.process()
    ctx.slot_boundary("FT8", 30)
    for ch, f, s in chans:
        iq = oracle.synth_iq(0xC0FFEE ^ s, n, FS, tones_hz=_tones(f), amp=2.0e4)
        oc = oracle.Channel("FT8", FS, IQ_LEN, f)
        oc.boundary(15)
        oc.push_many(iq); r = oc.boundary(30, want_f32=True)
        a, nv = ctx.fetch_audio_f32(ch)
        g = ctx.fetch_frame(ch)
        assert_frames_match(a, r["f32"])
        assert_int16_match(g["i16"], r["i16"], r["f32"] * r["factor"])


def test_channel_from_unchanged_decoder_line(ctx, oracle):
    """`decoder=28074000 FT8` on a receiver whose LO is 28.1 MHz tunes demod_hz = -26000 (Instance.cpp:183)."""
    n = 40 * IQ_LEN
    iq = oracle.synth_iq(8, n, FS, tones_hz=[-26000 + 1200.0], amp=1e4)
    rx = ctx.receiver_open(FS, IQ_LEN, 28100000)
    a = ctx.channel_open_line(rx, "28074000 FT8")
    b = ctx.channel_open(rx, -26000, "FT8")
    ctx.slot_boundary("FT8", 1); ctx.push_iq(rx, iq); ctx.slot_boundary("FT8", 2)
    fa, fb = ctx.fetch_frame(a), ctx.fetch_frame(b)
    assert np.array_equal(fa["i16"], fb["i16"]) and fa["i16"].any()
    ref = oracle.Channel("FT8", FS, IQ_LEN, -26000); ref.boundary(1); ref.push_many(iq)
    r = ref.boundary(2, want_f32=True)
    assert_int16_match(fa["i16"], r["i16"], r["f32"] * r["factor"])


def test_wav_file_is_the_reference_container(ctx, oracle, tmp_path):
    """BASELINE configs[0] asks for a 12 kHz .wav: header bytes per WaveFile.hpp:96-113, then the whole int16 frame."""
    n = 32 * IQ_LEN
    iq = oracle.synth_iq(12, n, FS, tones_hz=[-26000 + 1500.0], amp=1e4)
    rx = ctx.receiver_open(FS, IQ_LEN, 28100000)
    ch = ctx.channel_open_line(rx, "28074000 FT8")
    ctx.slot_boundary("FT8", 1); ctx.push_iq(rx, iq); ctx.slot_boundary("FT8", 2)
    p = tmp_path / "slot.wav"
    ctx.write_wav(ch, p)
    raw = p.read_bytes()
    fr = ctx.fetch_frame(ch)["i16"]
    assert raw[:46] == oracle.wav_header(240000) and len(raw) == 46 + 480000
    assert np.array_equal(np.frombuffer(raw[46:], np.int16), fr)


@pytest.mark.parametrize("seg", ["4", "12", "36", "100", "360", "1408"])
def test_exact5_stream_lengths(seg):
    """demod_exact5_kernel cuts a channel's pending outputs into waves of 32 streams x seg_len outputs; the launch picks seg_len from the amount of
    work.  Here it is forced (lab library: CWSLG_EXACT5_SEG_FORCE) through short, ragged and whole-slot values on the same 11 520-output slot of
    four channels: 90 / 30 / 10 / 4 / 1 / 1 waves per channel, the last one partly or mostly idle -- every frame must keep the reference's bits."""
    import os, subprocess, sys
    env = dict(os.environ, CWSLG_LIB="lab", CWSLG_EXACT5_SEG_FORCE=seg)
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "lab_variant_check.py"), "exact"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "lab check OK: demod_exact5_kernel" in r.stdout


@pytest.mark.parametrize("variant,mode", [("7", "fast"), ("8", "fast"), ("2", "fast"), ("20", "exact"), ("21", "exact"), ("23", "exact"), ("24", "exact"), ("27", "exact")])
def test_measured_alternative_kernels_live_in_the_lab_library(variant, mode):
    """The measured alternatives of the demod kernels (CWSLG_DEMOD_VARIANT: 7 = FIR on the f32 matrix cores, 8 = on the bf16 matrix
    cores with three-way split operands, 2 = persistent workgroups; exact mode: 20 / 21 = round 1's and round 2's kernels, 23 / 24 =
    demod_exact3_kernel with two-wave and one-wave workgroups, 27 = round 4's demod_exact4_kernel for every output) exist in libcwslgpu_lab.so only -- the product library has one kernel
    per job and reads no such switch -- and obey the mode's bound: 1e-5 of frame peak (fast), identical bits (exact).  Run in a
    child process: the library is chosen when the package is imported."""
    import os, subprocess, sys
    env = dict(os.environ, CWSLG_LIB="lab", CWSLG_DEMOD_VARIANT=variant)
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "lab_variant_check.py"), mode], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "lab check OK" in r.stdout


def test_ragged_launch_small_ring_beside_full_slot(ctx, oracle):
    """Two receivers of one sample rate in ONE launch: A has the smallest ring the library hands out and one block pending, B a ring of
    a whole slot with a whole slot pending.  The launch's tile count comes from B, so A's descriptor is drawn with tile indices far
    beyond its own range (n_out == 0): those items must touch neither A's small ring beyond its end nor A's checkpoint table beyond
    its length (round 3: the persistent kernels issued such an item's loads from the meaningless positions).  Both frames are
    checked against the oracle; the process must survive."""
    fa, fb = -26000, 40000
    nb = 1400 * IQ_LEN                                         # ~15 s: 5600 tiles of 512 outputs on B
    iq_a = oracle.synth_iq(5, 2 * IQ_LEN, FS, tones_hz=_tones(fa), amp=2.0e4)
    iq_b = oracle.synth_iq(6, IQ_LEN + nb, FS, tones_hz=_tones(fb), amp=2.0e4)
    rx_a = ctx.receiver_open(FS, IQ_LEN, 0, ring_blocks=1)     # clamped up to the minimum ring (two tiles + slack)
    rx_b = ctx.receiver_open(FS, IQ_LEN, 0, ring_blocks=nb // IQ_LEN + 8)
    cap_a = ctx.ring_info(rx_a)[1]
    assert cap_a < 40000, cap_a
    ch_a = ctx.channel_open(rx_a, fa, "FT8")
    ch_b = ctx.channel_open(rx_b, fb, "FT8")
    oa, ob = oracle.Channel("FT8", FS, IQ_LEN, fa), oracle.Channel("FT8", FS, IQ_LEN, fb)
    ctx.push_iq(rx_a, iq_a[:IQ_LEN]); oa.push_many(iq_a[:IQ_LEN])
    ctx.push_iq(rx_b, iq_b[:IQ_LEN]); ob.push_many(iq_b[:IQ_LEN])
    ctx.slot_boundary("FT8", 15); assert oa.boundary(15) is None and ob.boundary(15) is None
    ctx.push_iq(rx_a, iq_a[IQ_LEN:]); oa.push_many(iq_a[IQ_LEN:])          # one block pending on A ...
    for k in range(IQ_LEN, len(iq_b), 256 * IQ_LEN):                       # ... a whole slot on B (no process() in between)
        ctx.push_iq(rx_b, iq_b[k:k + 256 * IQ_LEN])
    ob.push_many(iq_b[IQ_LEN:])
    before = ctx.stats()["demod_launches"]
    ctx.process()                                                          # ONE launch for both
    assert ctx.stats()["demod_launches"] == before + 1
    ctx.slot_boundary("FT8", 30)
    ra, rb = oa.boundary(30, want_f32=True), ob.boundary(30, want_f32=True)
    for ch, r in ((ch_a, ra), (ch_b, rb)):
        a, nv = ctx.fetch_audio_f32(ch)
        g = ctx.fetch_frame(ch)
        assert nv == (IQ_LEN // 16 if ch == ch_a else nb // 16)
        assert_frames_match(a, r["f32"])
        assert_int16_match(g["i16"], r["i16"], r["f32"] * r["factor"])
