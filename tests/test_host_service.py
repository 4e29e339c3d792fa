"""Host-service rules either side of the hot path: slot clock, pool sizing, band lookup,
config.ini handling of the skimmer program.  Product = libcwslgpu.so's pure functions and cwsl_gpu_skimmer --dry-run
(no GPU needed); checker = oracle/host_oracle.c (the reference's loops run in virtual time)."""
import json
import os
import random
import subprocess

import numpy as np
import pytest

import cwsl_digi_amd as P
from cwsl_digi_amd import build as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T0 = 1_790_000_000_000          # an arbitrary UTC instant (ms), 2026


@pytest.mark.parametrize("group,span_s", [(0, 130), (1, 130), (2, 200), (3, 400), (4, 1500), (5, 2000), (6, 4000), (7, 8000)])
def test_slot_clock_matches_the_polling_threads(oracle, group, span_s):
    rng = random.Random(group)
    for trial in range(6):
        # start a little after a whole second that is NOT a boundary second, as a freshly started thread would
        start = T0 + rng.randrange(0, 3_600_000) // 1000 * 1000 + rng.randrange(30, 900)
        while P.slot_clock_next(group, start - 1000) <= start:
            start += 1000
        end = start + span_s * 1000
        fires = oracle.clock_sim(group, start, end)
        edges, e = [], P.slot_clock_next(group, start)
        while e < end - 500:
            edges.append(e)
            e = P.slot_clock_next(group, e)
        assert len(edges) >= 2 and len(fires) >= len(edges)
        # every computed boundary is what the thread fires on, seen at most one 25 ms poll late; nothing in between
        for k, e in enumerate(edges):
            assert 0 <= fires[k] - e <= 25, (group, k, e, fires[k])
        # the epoch Instance.cpp:214 stamps (whole seconds "now") equals edge // 1000
        assert all(f // 1000 == e // 1000 for f, e in zip(fires, edges))


def test_slot_clock_landmarks():
    t = 1_790_000_000_000 // 60000 * 60000                      # top of a minute
    assert P.slot_clock_next("FT8", t) == t + 15000 and P.slot_clock_next("FT8", t - 1) == t
    assert [P.slot_clock_next("FT4", t + d) - t for d in (0, 7399, 7400, 15000, 22399)] == [7400, 7400, 15000, 22400, 22400]
    assert P.slot_clock_next("Q65-30", t + 1) == t + 30000
    assert P.slot_clock_next("JT65", t) == t + 60000
    hour = t // 3_600_000 * 3_600_000
    assert P.slot_clock_next("WSPR", hour + 1) == hour + 120000
    assert P.slot_clock_next("FST4W-300", hour + 1) == hour + 300000
    assert P.slot_clock_next("FST4-900", hour + 1) == hour + 900000
    assert P.slot_clock_next("FST4W-1800", hour + 1) == hour + 1800000
    assert P.slot_clock_next(99, t) == 0


def test_pool_sizing_matches(oracle):
    rng = random.Random(7)
    for _ in range(400):
        counts = [rng.randrange(0, 40) if rng.random() < 0.6 else 0 for _ in range(8)]
        if sum(counts) == 0:
            counts[1] = 1
        burden = rng.choice([1.0, 0.5, 1.5, 2.0, 0.75, 3.3])
        assert P.pool_sizing(counts, burden) == oracle.pool_sizing(counts, burden), (counts, burden)
    # spelled out: 20 FT8 + 10 FT4 -> 6 + 0.55 -> 7 ; 3 WSPR among 33 decoders -> round(7.55->8 ...) etc.
    assert P.pool_sizing([10, 20, 0, 0, 0, 0, 0, 0]) == (7, 0)
    assert P.pool_sizing([0, 1, 0, 0, 1, 0, 0, 0]) == (1, 1)


def test_find_band_matches(oracle):
    bands = [(14_100_000, 192000), (7_100_000, 96000), (14_150_000, 192000), (3_550_000, 48000)]
    for f in (14_074_000, 14_196_000, 14_196_001, 14_004_000, 14_003_999, 7_074_000, 7_148_000, 7_148_001, 3_573_000, 28_074_000, 14_200_000):
        assert P.find_band(bands, f) == oracle.find_band(bands, f), f
    assert P.find_band(bands, 14_074_000) == 0 and P.find_band(bands, 14_200_000) == 2 and P.find_band(bands, 28_074_000) == -1


CONFIG = """
# comment
[radio]
freqcalibration=1.00000250
sharedmem=2
[operator]
callsign=N0CALL
gridsquare=AB01xy
[decoders]
decoder=14074000 FT8
decoder=14080000 FT4
decoder=14095600 WSPR 2 1.0000010 K1ABC
decoder=7074000 FT8
decoder=7047500 FST4W-120
[wsjtx]
temppath=/tmp/wav
binpath=/opt/wsjtx/bin
highestdecodefreq=9000
decodedepth=7
numjt9threads=0
wsprcycles=500
transfermethod=shmem
ftaudioscalefactor=0.8
"""


def _skimmer():
    return B.build_skimmer()


def test_skimmer_dry_run_reads_config_like_the_reference(tmp_path, oracle):
    cfg = tmp_path / "config.ini"
    cfg.write_text(CONFIG)
    a = tmp_path / "a.c64"; a.write_bytes(b"")
    out = subprocess.run([_skimmer(), "--config", str(cfg), "--rx", f"file={a},fs=192000,block=2048,lo=14100000",
                          "--rx", f"file={a},fs=96000,block=1024,lo=7060000", "--dry-run"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout)
    assert d["decoders"] == 5 and d["receivers"] == 2
    assert d["highestdecodefreq"] == 6000 and d["decodedepth"] == 3 and d["numjt9threads"] == 1     # clamps :893, :942, :1017
    assert "wsjtx.decodedepth is too high, setting to 3" in out.stderr and "wsjtx.numjt9threads is too small, setting to 1" in out.stderr
    assert d["wsprcycles"] == 500 and d["transfer_shmem"] == 1 and abs(d["ft_scale"] - 0.8) < 1e-7 and abs(d["wspr_scale"] - 0.2) < 1e-7
    counts = [1, 2, 0, 0, 1, 0, 1, 0]
    assert (d["numjt9instances"], d["maxwsprdinstances"]) == oracle.pool_sizing(counts, 1.0, 5)
    plan = d["plan"]
    assert [p["rx"] for p in plan] == [0, 0, 0, 1, 1]
    assert [p["route"] for p in plan] == ["shmem", "shmem", "wavefile", "shmem", "wavefile"]        # DecoderPool.hpp:379-395
    spec = P.parse_decoder_line("14095600 WSPR 2 1.0000010 K1ABC", 1.00000250)
    assert plan[2]["calibrated_hz"] == spec["calibrated_hz"] and plan[2]["demod_hz"] == spec["calibrated_hz"] - 14100000
    assert plan[0]["demod_hz"] == P.parse_decoder_line("14074000 FT8", 1.00000250)["calibrated_hz"] - 14100000


@pytest.mark.parametrize("line,msg", [
    ("ftaudioscalefactor=1.5", "ftaudioscalefactor must be <= 1.0"),
    ("wspraudioscalefactor=0", "wsjtx.wspraudioscalefactor must be > 0"),
    ("wsprcycles=20000", "wsjtx.wsprcycles must be <= 10000"),
    ("numjt9instances=0", "wsjtx.numjt9instances must be >= 1"),
])
def test_skimmer_config_fatal_errors_use_the_reference_messages(tmp_path, line, msg):
    cfg = tmp_path / "config.ini"
    cfg.write_text("[decoders]\ndecoder=14074000 FT8\n[wsjtx]\n" + line + "\n")
    a = tmp_path / "a.c64"; a.write_bytes(b"")
    out = subprocess.run([_skimmer(), "--config", str(cfg), "--rx", f"file={a},fs=192000,block=2048,lo=14100000", "--dry-run"],
                         capture_output=True, text=True)
    assert out.returncode == 1 and msg in out.stderr


def test_skimmer_rejects_missing_decoders_and_uncovered_bands(tmp_path):
    a = tmp_path / "a.c64"; a.write_bytes(b"")
    cfg = tmp_path / "c1.ini"; cfg.write_text("[wsjtx]\ndecodedepth=2\n")
    out = subprocess.run([_skimmer(), "--config", str(cfg), "--rx", f"file={a},fs=192000,block=2048,lo=14100000", "--dry-run"], capture_output=True, text=True)
    assert out.returncode == 1 and "decoders.decoder input is required but was not specified!" in out.stderr
    cfg.write_text("[decoders]\ndecoder=28074000 FT8\n")
    out = subprocess.run([_skimmer(), "--config", str(cfg), "--rx", f"file={a},fs=192000,block=2048,lo=14100000", "--dry-run"], capture_output=True, text=True)
    assert out.returncode == 1 and "no receiver covers it" in out.stderr


def test_skimmer_shards_decoders_by_receiver(tmp_path):
    """One process per GPU: receiver k and every decoder on it belong to rank k mod world (a band's IQ goes to
    exactly one GPU; the reference creates one Receiver per band, CWSL_DIGI.cpp:115-129).  --dry-run prints the rank's share."""
    cfg = tmp_path / "config.ini"
    cfg.write_text(CONFIG)
    a = tmp_path / "a.c64"; a.write_bytes(b"")
    base = [_skimmer(), "--config", str(cfg), "--rx", f"file={a},fs=192000,block=2048,lo=14100000",
            "--rx", f"file={a},fs=96000,block=1024,lo=7060000", "--dry-run"]
    whole = json.loads(subprocess.run(base, capture_output=True, text=True).stdout)["plan"]
    shares = []
    for rank in range(2):
        out = subprocess.run(base + ["--world", "2", "--rank", str(rank), "--rccl-id", str(tmp_path / "id"), "--start-ms", "1700000000000"],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        plan = json.loads(out.stdout)["plan"]
        assert plan and all(p["rx"] % 2 == rank for p in plan)
        shares += plan
    assert sorted(shares, key=lambda p: (p["rx"], p["freq_hz"])) == sorted(whole, key=lambda p: (p["rx"], p["freq_hz"]))
    # a rank outside the world, no id file, or no common --start-ms (the ranks must fire the same boundary sequence): usage errors
    for bad in (["--world", "2", "--rank", "2", "--rccl-id", "x", "--start-ms", "1"], ["--world", "2", "--rank", "0", "--start-ms", "1"],
                ["--world", "2", "--rank", "0", "--rccl-id", "x"]):
        assert subprocess.run(base + bad, capture_output=True, text=True).returncode == 2
