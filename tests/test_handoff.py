"""Decoder hand-off formats: jt9/js8 shared-memory block layout and fill, decoder command lines.
CPU tests compare the product's host logic (through the C ABI; no GPU call) with oracle/handoff_oracle.c; the GPU
test fills a block from a real finalised frame and requires it byte-identical to the oracle's block."""
import numpy as np
import pytest

import cwsl_digi_amd as P

MODES = ["FT8", "FT4", "JS8", "WSPR", "Q65-30", "JT65", "FST4-60", "FST4-120", "FST4-300", "FST4-900", "FST4-1800",
         "FST4W-120", "FST4W-300", "FST4W-900", "FST4W-1800"]
JT9_FIELDS = ["nutc", "ndiskdat", "ntrperiod", "nQSOProgress", "nfqso", "nftx", "newdat", "npts8", "nfa", "nfSplit", "nfb",
              "ntol", "kin", "nzhsym", "nsubmode", "nagain", "ndepth", "lft8apon", "lapcqonly", "ljt65apon", "napwid",
              "ntxmode", "nmode", "minw", "nclearave", "minSync", "emedelay", "dttol", "nlist", "listutc", "n2pass",
              "nranera", "naggressive", "nrobust", "nexp_decode", "datetime", "mycall", "mygrid", "hiscall", "hisgrid"]
JS8_EXTRA = ["syncStats", "kposA", "kposB", "kposC", "kposE", "kposI", "kszA", "kszB", "kszC", "kszE", "kszI", "nsubmodes",
             "ndebug"]


def test_block_sizes_and_offsets_match_the_c_structs(oracle):
    for js8 in (False, True):
        assert P.decoder_block_bytes(js8) == oracle.decoder_block_bytes(js8)
        names = ["ss", "savg", "sred", "d2", "params"] + [f for f in JT9_FIELDS if not (js8 and f == "nfSplit")]
        names += JS8_EXTRA if js8 else ["ipc"]
        for n in names:
            assert P.decoder_block_field(n, js8)[0] == oracle.decoder_block_offset(n, js8), (js8, n)
    # landmarks: ipc[3] leads the jt9 block only; d2 holds 30 minutes of 12 kHz int16
    assert P.decoder_block_field("ss")[0] == 12 and P.decoder_block_field("ss", True)[0] == 0
    assert P.decoder_block_field("d2")[1] == 30 * 60 * 12000 * 2
    with pytest.raises(P.CwslGpuError):
        P.decoder_block_field("nfSplit", True)
    with pytest.raises(P.CwslGpuError):
        P.decoder_block_field("ipc", True)


def test_route_matches_decoderpool_dispatch(oracle):
    for m in MODES:
        for shm in (True, False):
            assert P.decoder_route(m, shm) == oracle.decoder_route(m, shm), m
    assert P.decoder_route("FT8") == "shmem" and P.decoder_route("FT8", False) == "wavefile"
    assert [P.decoder_route(m) for m in ("WSPR", "JS8", "FST4-120", "FST4W-300")] == ["wavefile"] * 4
    with pytest.raises(P.CwslGpuError):
        P.decoder_route("PSK31")


def test_command_lines_character_for_character(oracle):
    for m in MODES:
        for shm in (True, False):
            for th, dp, hz, cyc, per in ((3, 3, 3000, 3000, 120.0), (1, 1, 6000, 100, 300.0), (9, 2, 2500, 10000, 1800.0)):
                want = oracle.decoder_command(m, "KEY-or-file.wav", shm, th, dp, hz, cyc, per)
                if want is None:
                    with pytest.raises(P.CwslGpuError):
                        P.decoder_command(m, "KEY-or-file.wav", shm, th, dp, hz, cyc, per)
                else:
                    assert P.decoder_command(m, "KEY-or-file.wav", shm, th, dp, hz, cyc, per) == want, (m, shm)
    # known answers spelled out from the reference's concatenations (DecoderPool.hpp:634-659, 1007-1046)
    assert P.decoder_command("FT8", "abc", True) == ("jt9.exe", " -8 -m 3  -s abc")
    assert P.decoder_command("FT4", "C:\\t\\x.wav", False, 2, 1, 2800) == ("jt9.exe", " -5 -m 2 -d 1 -w 1 -H 2800 C:\\t\\x.wav")
    assert P.decoder_command("WSPR", "w.wav", False, wspr_cycles=500) == ("wsprd.exe", " -C 500 -o 5 -d w.wav")
    assert P.decoder_command("FST4W-300", "w.wav", False, trperiod=300.0) == ("jt9.exe", " -W -p 300 -m 3 -d 3 -L 1400 -H 1600 -F 200 w.wav")
    assert P.decoder_command("JS8", "w.wav", False) == ("js8.exe", " -8 -m 3 w.wav")
    assert P.decoder_command("Q65-30", "k", True, highest_decode_hz=2900) == ("jt9.exe", " -3 -m 3 -p 30 -H 2900  -s k")


@pytest.mark.gpu
@pytest.mark.parametrize("mode,js8", [("FT8", False), ("FT4", False), ("JT65", False), ("Q65-30", False), ("FST4W-120", False),
                                      ("FST4-60", False), ("JS8", True)])
def test_gpu_block_is_byte_identical_to_oracle(ctx, oracle, mode, js8):
    fs, blk = 48000, 1024
    rx = ctx.receiver_open(fs, blk, 7000000)
    ch = ctx.channel_open(rx, 3000, mode)
    grp = mode
    n = 40 * blk
    iq = oracle.synth_iq(77, n, fs, tones_hz=[3000 + 1500.0], amp=1.2e4)
    assert ctx.fill_decoder_block(ch, js8=js8) is None                    # nothing finalised yet
    ctx.slot_boundary(grp, 500); ctx.push_iq(rx, iq); ctx.slot_boundary(grp, 560)
    fr = ctx.fetch_frame(ch)
    got, t0 = ctx.fill_decoder_block(ch, js8=js8, decodedepth=2, highest_decode_hz=2900)
    want = oracle.decoder_block(mode, fr["i16"], 2, 2900, js8=js8)
    assert t0 == 500 and got.nbytes == want.nbytes
    assert np.array_equal(got, want)
    off, nbytes = P.decoder_block_field("d2", js8)
    d2 = got[off:off + 2 * len(fr["i16"])].view(np.int16)
    assert np.array_equal(d2, fr["i16"]) and np.any(d2 != 0)


@pytest.mark.gpu
def test_gpu_block_rejects_modes_and_small_buffers(ctx, oracle):
    rx = ctx.receiver_open(48000, 1024, 0)
    ch = ctx.channel_open(rx, 0, "WSPR")
    iq = oracle.synth_iq(5, 8 * 1024, 48000)
    ctx.slot_boundary("WSPR", 1); ctx.push_iq(rx, iq); ctx.slot_boundary("WSPR", 2)
    with pytest.raises(P.CwslGpuError) as e:
        ctx.fill_decoder_block(ch)                                         # DecoderPool.hpp:566-570 "Unknown mode"
    assert e.value.status == -5
    with pytest.raises(P.CwslGpuError):
        ctx.fill_decoder_block(ch, js8=True)                               # js8 block only for JS8 channels
    with pytest.raises(P.CwslGpuError):
        ctx.fill_decoder_block(ch, block=np.empty(1000, np.uint8))
