"""GPU: the runnable host program (cwsl_gpu_skimmer) end to end -- config.ini + band files in, slot clock from the
sample count, .wav / candidate files out -- against the oracle driven with the same blocks and boundaries."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

import cwsl_digi_amd as P
from cwsl_digi_amd import build as B
from ft8_signal import ft8_iq, ft4_iq

pytestmark = pytest.mark.gpu

CONFIG = """
[radio]
freqcalibration=1.0000000
[decoders]
decoder=14074000 FT8
decoder=14080000 FT4
decoder=7074000 FT8
decoder=7078000 JS8
[wsjtx]
binpath=/opt/wsjtx/bin
highestdecodefreq=3000
transfermethod=shmem
"""


def _read_wav(path):
    raw = open(path, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt " and raw[38:42] == b"data"
    return raw[:46], np.frombuffer(raw[46:], np.int16)


def test_skimmer_files_match_oracle(tmp_path, oracle):
    rng = np.random.default_rng(5)
    dur = 34.0
    rxs = [dict(fs=192000, block=2048, lo=14_100_000, header=True), dict(fs=96000, block=1024, lo=7_060_000, header=False)]
    decs = [("FT8", 14_074_000, 0), ("FT4", 14_080_000, 0), ("FT8", 7_074_000, 1), ("JS8", 7_078_000, 1)]
    start_ms = 1_790_000_000_000 // 15000 * 15000 + 12_000           # 3 s before an FT8/FT4 boundary
    iqs = []
    for k, r in enumerate(rxs):
        n = int(dur * r["fs"]) // r["block"] * r["block"]
        iq = oracle.synth_iq(100 + k, n, r["fs"], tones_hz=[], amp=0.0) * 0.05
        if k == 0:
            iq = iq + ft8_iq(r["fs"], n, -26000, 1200.0, 3.6, 3000.0, rng) + ft8_iq(r["fs"], n, -26000, 1900.0, 18.9, 2500.0, rng)
            iq = iq + ft4_iq(r["fs"], n, -20000, 1000.0, 3.4, 3000.0, rng) + ft4_iq(r["fs"], n, -20000, 1500.0, 10.8, 3000.0, rng)
        else:
            iq = iq + ft8_iq(r["fs"], n, 14000, 800.0, 3.7, 2800.0, rng) + ft8_iq(r["fs"], n, 18000, 1000.0, 3.5, 2800.0, rng)
        iq = iq.astype(np.complex64)
        path = tmp_path / f"band{k}.iq"
        with open(path, "wb") as f:
            if r["header"]:
                f.write(struct.pack("<iii", r["fs"], r["block"], r["lo"]))     # SM_HDR (SharedMemory.h:10-21)
            iq.tofile(f)
        r["path"] = path
        iqs.append(iq)
    cfg = tmp_path / "config.ini"
    cfg.write_text(CONFIG)
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    args = [B.build_skimmer(), "--config", str(cfg), "--out", str(out_dir), "--start-ms", str(start_ms), "--exact", "--wav", "always",
            "--rx", f"file={rxs[0]['path']},header=1", "--rx", f"file={rxs[1]['path']},fs=96000,block=1024,lo=7060000"]
    run = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout, run.stderr)
    summary = json.loads(run.stdout.strip().splitlines()[-1])

    # ---- the same schedule on the oracle
    chans = [oracle.Channel(m, rxs[r]["fs"], rxs[r]["block"], f - rxs[r]["lo"]) for m, f, r in decs]
    groups = sorted({P.group_of(m) for m, _, _ in decs})
    nxt = {g: P.slot_clock_next(g, start_ms) for g in groups}
    pos = [0, 0]
    expect = {}
    n_bound = 0
    while True:
        live = [k for k in range(2) if pos[k] < len(iqs[k])]
        if not live:
            break
        k = min(live, key=lambda q: (pos[q] / rxs[q]["fs"], q))
        blk = iqs[k][pos[k]:pos[k] + rxs[k]["block"]]
        for c, (m, f, r) in zip(chans, decs):
            if r == k:
                c.push(blk)
        pos[k] += rxs[k]["block"]
        live = [q for q in range(2) if pos[q] < len(iqs[q])]
        t = min(pos[q] / rxs[q]["fs"] for q in live) if live else pos[k] / rxs[k]["fs"]
        now = start_ms + int(t * 1000.0)
        for g in groups:
            while nxt[g] <= now:
                n_bound += 1
                for c, (m, f, r) in zip(chans, decs):
                    if P.group_of(m) == g:
                        fr = c.boundary(nxt[g] // 1000)
                        if fr is not None:
                            expect[(fr["t_start"], f, m)] = fr
                nxt[g] = P.slot_clock_next(g, nxt[g])
    assert summary["boundaries"] == n_bound and summary["frames"] == len(expect) and summary["blocks_dropped"] == 0
    assert len(expect) >= 2 + 4 + 2 + 2 - 4                                   # every decoder produced complete slots
    log = [json.loads(l) for l in open(out_dir / "frames.jsonl")]
    assert len(log) == len(expect)
    seen_cands = 0
    for (t0, f, m), fr in expect.items():
        hdr, pcm = _read_wav(out_dir / f"{t0}_{f}_{m}.wav")
        assert hdr == oracle.wav_header(len(fr["i16"]))
        assert np.array_equal(pcm, fr["i16"]), (t0, f, m)                     # exact mode: bit-identical files
        rec = [r for r in log if (r["t_start"], r["freq_hz"], r["mode"]) == (t0, f, m)][0]
        route = P.decoder_route(m, True)
        assert rec["route"] == route
        want_app, want_opts = oracle.decoder_command(m, "<shmem-key>" if route == "shmem" else rec["wav"], route == "shmem")
        assert (rec["app"], rec["opts"]) == (want_app, want_opts)
        if m in ("FT8", "FT4"):
            want = oracle.ft8_sync(fr["i16"], 200, 3000, 1.5, 200) if m == "FT8" else oracle.ft4_candidates(fr["i16"], 200.0, 3000.0, 1.2, 200)
            got = [tuple(float(x) for x in l.split()) for l in open(out_dir / f"{t0}_{f}_{m}.cand")]
            assert len(got) == len(want) == rec["candidates"]
            for g_, w_ in zip(got, want):
                assert tuple(np.float32(x) for x in g_) == (np.float32(w_[3]), np.float32(w_[4]), np.float32(w_[2]))      # %.9g round-trips float32
            seen_cands += len(want)
            if m == "FT4":                                                    # coherent refinement records
                ref4 = oracle.ft4_sync_all(fr["i16"], want)
                got4 = [l.split() for l in open(out_dir / f"{t0}_{f}_{m}.sync4")]
                assert len(got4) == len(ref4) == rec["ft4_refined"]
                for g_, w_ in zip(got4, ref4):
                    assert tuple(np.float32(x) for x in g_[:4]) == tuple(np.float32(w_[k]) for k in ("f0_hz", "f1_hz", "dt_s", "sync"))
                    assert [int(x) for x in g_[4:]] == [w_["ibest"], w_["idf"], w_["seg"], w_["cand"]]
    assert seen_cands > 0


def test_skimmer_udp_source(tmp_path, oracle):
    """One receiver fed by UDP datagrams on loopback: frames equal the file-fed oracle schedule."""
    import socket, threading, time
    fs, block, lo = 48000, 1024, 3_560_000
    n = int(20.0 * fs) // block * block
    iq = oracle.synth_iq(9, n, fs, tones_hz=[13000 + 1500.0], amp=1.0e4).astype(np.complex64)
    cfg = tmp_path / "config.ini"
    cfg.write_text("[decoders]\ndecoder=3573000 FT8\n")
    start_ms = 1_790_000_000_000 // 15000 * 15000 + 13_000
    summary = None
    for attempt in range(4):                                                  # loopback may drop datagrams under load: retry with slower pacing, then FAIL
        out_dir = tmp_path / f"out{attempt}"; out_dir.mkdir()
        port = 47000 + (os.getpid() + 17 * attempt) % 1000
        proc = subprocess.Popen([B.build_skimmer(), "--config", str(cfg), "--out", str(out_dir), "--start-ms", str(start_ms), "--exact",
                                 "--wav", "always", "--rx", f"udp={port},fs={fs},block={block},lo={lo},idle=2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        first = proc.stderr.readline()                                        # "ready: ..." once the context exists and the port is bound
        assert first.startswith("ready"), first + proc.stderr.read()
        s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        every = max(1, 4 >> attempt)
        for k in range(n // block):
            s.sendto(iq[k * block:(k + 1) * block].tobytes(), ("127.0.0.1", port))
            if k % every == every - 1:
                time.sleep(0.001 * (1 + attempt))                             # keep the loopback socket buffer from overflowing
        s.close()
        out, err = proc.communicate(timeout=120)                              # the 2 s receive timeout ends the run
        assert proc.returncode == 0, (out, err)
        summary = json.loads(out.strip().splitlines()[-1])
        if summary["pushed_samples"] == n:
            break
    assert summary["pushed_samples"] == n, f"loopback dropped datagrams in four attempts: {summary}"
    c = oracle.Channel("FT8", fs, block, 3_573_000 - lo)
    nxt = P.slot_clock_next("FT8", start_ms)
    expect = []
    for k in range(n // block):
        c.push(iq[k * block:(k + 1) * block])
        now = start_ms + int((k + 1) * block / fs * 1000.0)
        while nxt <= now:
            fr = c.boundary(nxt // 1000)
            if fr is not None:
                expect.append(fr)
            nxt = P.slot_clock_next("FT8", nxt)
    assert summary["frames"] == len(expect) == 1
    _, pcm = _read_wav(out_dir / f"{expect[0]['t_start']}_3573000_FT8.wav")
    assert np.array_equal(pcm, expect[0]["i16"])
