"""Decoder stdout -> spot record for FT8/FT4: the product's parser (C ABI, no GPU) against
oracle/spot_oracle.c and against answers written out by hand from OutputHandler.cpp:505-621, 924-1128."""
import random

import pytest

import cwsl_digi_amd as P


def L(msg, snr=-12, dt=0.3, freq=1234, flag="~", t="123045"):
    """A jt9 -8 / -5 stdout line in its fixed columns."""
    return f"{t} {snr:>3} {dt:>4.1f} {freq:>4} {flag}  {msg}"


KNOWN = [
    # (mode, message, status, call, locator)
    ("FT8", "CQ K1ABC FN42", "ok", "K1ABC", "FN42"),
    ("FT8", "CQ K1ABC", "ok", "K1ABC", None),
    ("FT8", "CQ DX K1ABC FN42", "ok", "K1ABC", "FN42"),
    ("FT8", "CQ POTA W9XYZ", "ok", "W9XYZ", None),                 # CQ SOMETHING CALL: first token fails checkCall
    ("FT8", "CQ K1ABC QRP", "ok", "K1ABC", None),                  # CQ CALL SOMETHING
    ("FT8", "CQ NA K1ABC XX", "unhandled", "", None),              # 3 spaces needs a valid grid
    ("FT8", "W2AXR K1ABC -07", "ok", "K1ABC", None),
    ("FT8", "W2AXR K1ABC RR73", "ok", "K1ABC", None),
    ("FT8", "W2AXR K1ABC R FN42", "ok", "K1ABC", "FN42"),
    ("FT8", "W2AXR K1ABC R XX42", "ok", "K1ABC", "XX42"),
    ("FT8", "N4ZR W2AXR 599 NY", "ok", "W2AXR", None),
    ("FT8", "N4ZR W2AXR 599 0244", "ok", "W2AXR", None),           # RST + serial (the reference's own example)
    ("FT8", "N4ZR W2AXR 59 0244", "unhandled", "", None),          # third token must be 3 wide (spaces 4 apart)
    ("FT8", "<PJ4/K1ABC> W9XYZ", "ok", "W9XYZ", None),
    ("FT8", "TNX BOB 73", "unhandled", "", None),                  # middle token is all letters
    ("FT8", "SOTAMAT K1ABC/AB", "unhandled", "", None),            # length rule: 7 + 8 + 1 != 13
    ("FT8", "STM K1ABC/12A", "ok", "K1ABC/12A", None),             # 3 + 9 + 1 == 13
    ("FT8", "K1ABC RR73; W9XYZ <KH1/KH7Z> -08", "ok", "KH1/KH7Z", None),    # Fox/Hound: second part, 2 spaces -> middle token
    ("FT4", "CQ K1ABC FN42", "ok", "K1ABC", "FN42"),
    ("FT4", "K1ABC RR73; W9XYZ <KH1/KH7Z> -08", "unhandled", "", None),     # no F/H split in FT4: 5 spaces
    ("FT8", "CQ K1ABC FN42 ? a1", "ok", "K1ABC", "FN42"),          # error flags chopped
    ("FT8", "W2AXR K1ABC 73 a2", "ok", "K1ABC", None),
    ("FT8", "CQ FN42", "unhandled", "", None),                     # a grid is not a call
    ("FT8", "CQ 12345", "unhandled", "", None),
    ("FT8", "73", "skip", "", None),                               # whole line <= 28 characters (OutputHandler.cpp:521)
    ("FT8", "TU 73", "unhandled", "", None),                       # shorter than 6 after trimming
]


@pytest.mark.parametrize("mode,msg,status,call,loc", KNOWN)
def test_known_answers(oracle, mode, msg, status, call, loc):
    line = L(msg, snr=-7, dt=-0.4, freq=987)
    got = P.parse_decode_line(mode, line, 14074000)
    assert got == oracle.parse_decode_line(mode, line, 14074000)
    assert (got["status"], got["call"], got["locator"]) == (status, call, loc), got
    if status != "skip":
        assert got["snr_db"] == -7 and abs(got["dt_s"] + 0.4) < 1e-6 and got["freq_hz"] == 14074987


def test_line_grammar(oracle):
    good = L("CQ K1ABC FN42")
    assert P.parse_decode_line("FT8", good, 7074000)["status"] == "ok"
    assert P.parse_decode_line("FT4", L("CQ K1ABC FN42", flag="+"), 7047500)["status"] == "ok"
    bad = ["<DecodeFinished>   0   3        0", "000000  -1  0.1 1500 ~", good[:6] + "x" + good[7:], good[:10] + "x" + good[11:],
           good[:15] + "x" + good[16:], good[:20] + "x" + good[21:], good[:21] + "#" + good[22:], good[:22] + "x" + good[23:],
           good[:23] + "x" + good[24:], "", "   ", "123045 abc  0.3 1234 ~  CQ K1ABC FN42"]
    for b in bad:
        assert P.parse_decode_line("FT8", b, 0)["status"] == "skip", b
        assert oracle.parse_decode_line("FT8", b, 0)["status"] == "skip", b
    with pytest.raises(P.CwslGpuError):
        P.parse_decode_line("JS8", good, 0)
    # leading / trailing blanks are trimmed before the columns are read (OutputHandler.cpp:515)
    assert P.parse_decode_line("FT8", "   " + good + "  \r", 0)["call"] == "K1ABC"


def test_random_messages_agree_with_oracle(oracle):
    rng = random.Random(11)
    toks = ["CQ", "K1ABC", "W9XYZ", "<PJ4/K1ABC>", "FN42", "RR73", "73", "-12", "R", "R-07", "599", "NY", "DX", "POTA", "QRP",
            "0244", "SM", "K1ABC/AB", "JA1XYZ/P", "?", "a1", "q3", "<...>", "TU;", "EA8/DL1ABC", "5B4AMM", "A", "1234", "G4"]
    n_ok = 0
    for _ in range(3000):
        msg = " ".join(rng.choice(toks) for _ in range(rng.randrange(1, 6)))
        for mode in ("FT8", "FT4"):
            line = L(msg, snr=rng.randrange(-24, 20), dt=rng.randrange(-20, 20) / 10, freq=rng.randrange(200, 3000))
            got, want = P.parse_decode_line(mode, line, 10136000), oracle.parse_decode_line(mode, line, 10136000)
            assert got == want, (mode, msg, got, want)
            n_ok += got["status"] == "ok"
    assert n_ok > 200


# ---- WSPR (wsprd), FST4W and FST4 (jt9 -W / -7) lines: OutputHandler.cpp:314-402, 152-240, 243-312
TOKEN_KNOWN = [
    # (mode, line, base, status, call, locator, freq_hz, snr, drift, dbm)
    ("WSPR", "9550  -0  0.3   0.001549  0  W8EDU EN91 23", 14095600, "ok", "W8EDU", "EN91", 14097149, 0, 0, 23),
    ("WSPR", "6a80 -20  0.2   0.001478  -1 <G0FCA> IO83UQ 30", 7038600, "ok", "G0FCA", "IO83UQ", 7040078, -20, -1, 30),
    ("WSPR", "4dab -25 -0.1", 14095600, "skip", "", None, 0, 0, 0, 0),
    ("WSPR", "9550   8  0.1   0.001574  0  QRP FN20 33", 14095600, "unhandled", "", None, 14097174, 8, 0, 33),
    ("FST4W-120", "0000 -13  0.4 1480 `  W3TS FN10 30", 474200, "ok", "W3TS", "FN10", 475680, -13, 0, 30),
    ("FST4W-300", "0000 -28  0.1 1512 `  <PJ4/K1ABC> FK52 23", 136000, "ok", "<PJ4/K1ABC>", "FK52", 137512, -28, 0, 23),   # no unpacking here
    ("FST4W-120", "0000 -13  0.4 1480 ~  W3TS FN10 30", 474200, "skip", "", None, 0, 0, 0, 0),
    ("FST4-60", "0000 -13  0.4 1080 `  CQ W3TS FN10", 1836600, "ok", "W3TS", "FN10", 1837680, -13, 0, 0),
    ("FST4-120", "0000  -7 -0.2  990 `  K1ABC W9XYZ R-05", 1836600, "ok", "W9XYZ", None, 1837590, -7, 0, 0),
    ("FST4-300", "0000  -7 -0.2  990 `  TNX 73 GL", 1836600, "unhandled", "", None, 1837590, -7, 0, 0),
]


@pytest.mark.parametrize("mode,line,base,status,call,loc,freq,snr,drift,dbm", TOKEN_KNOWN)
def test_token_modes_known_answers(oracle, mode, line, base, status, call, loc, freq, snr, drift, dbm):
    got = P.parse_decode_line(mode, line, base)
    assert got == oracle.parse_decode_line(mode, line, base)
    assert (got["status"], got["call"], got["locator"]) == (status, call, loc), got
    if status != "skip":
        assert (got["freq_hz"], got["snr_db"], got["drift"], got["dbm"]) == (freq, snr, drift, dbm), got


def test_token_modes_random_lines_agree_with_oracle(oracle):
    rng = random.Random(21)
    calls = ["W8EDU", "<G0FCA>", "QRP", "K1ABC/P", "FN20", "5B4AMM", "A1", "<...>"]
    locs = ["EN91", "IO83UQ", "FN", "XX00aa", "1234"]
    for _ in range(1500):
        mode = rng.choice(["WSPR", "FST4W-120", "FST4W-1800", "FST4-60", "FST4-900"])
        snr, dt, f = rng.randrange(-33, 12), rng.randrange(-15, 30) / 10, rng.randrange(900, 1600)
        if mode == "WSPR":
            toks = [f"{rng.randrange(0, 65536):04x}", f"{snr:>3}", f"{dt:>4.1f}", f"{f / 1e6:>10.6f}", f"{rng.randrange(-3, 4):>2}",
                    rng.choice(calls), rng.choice(locs), str(rng.choice([0, 10, 23, 30, 37]))]
            if rng.random() < 0.1:
                toks = toks[:rng.randrange(1, 8)]
            line = "  ".join(toks)
        elif mode.startswith("FST4W"):
            line = f"0000 {snr:>3} {dt:>4.1f} {f:>4} {rng.choice('`~')}  {rng.choice(calls)} {rng.choice(locs)} {rng.choice([23, 30])}"
        else:
            line = f"0000 {snr:>3} {dt:>4.1f} {f:>4} `  " + " ".join(rng.choice(["CQ", "K1ABC", "W9XYZ", "FN42", "R-05", "73", "RR73"]) for _ in range(rng.randrange(1, 5)))
        assert P.parse_decode_line(mode, line, 474200) == oracle.parse_decode_line(mode, line, 474200), (mode, line)
    for m in ("JS8", "PSK31"):
        with pytest.raises(P.CwslGpuError):
            P.parse_decode_line(m, "x", 0)


def test_jt65_and_q65_columns(oracle):
    """JT65: "HHMM snr  dt freq #  msg" (OutputHandler.cpp:623-695); Q65: FT8's columns without the Fox/Hound split (:697-780)."""
    jt = "0001 -11  0.3 1234 #  CQ K1ABC FN42"
    assert jt[4] == " " and jt[8] == " " and jt[13] == " " and jt[19] == "#" and jt[20] == " " and jt[22:] == "CQ K1ABC FN42"
    for line, mode in ((jt, "JT65"), ("123045 -12  0.3 1234 ~  CQ K1ABC FN42", "Q65-30"), ("123045 -12  0.3 1234 ~  K1ABC RR73; W9XYZ <KH1/KH7Z> -08", "Q65-30")):
        got = P.parse_decode_line(mode, line, 14076000)
        assert got == oracle.parse_decode_line(mode, line, 14076000)
    got = P.parse_decode_line("JT65", jt, 14076000)
    assert (got["status"], got["call"], got["locator"], got["freq_hz"], got["snr_db"]) == ("ok", "K1ABC", "FN42", 14077234, -11)
    assert P.parse_decode_line("Q65-30", "123045 -12  0.3 1234 ~  K1ABC RR73; W9XYZ <KH1/KH7Z> -08", 0)["status"] == "unhandled"
    assert P.parse_decode_line("JT65", jt[:4] + "x" + jt[5:], 0)["status"] == "skip"
    rng = random.Random(31)
    for _ in range(1000):
        msg = " ".join(rng.choice(["CQ", "K1ABC", "W9XYZ", "FN42", "R-05", "73", "RR73", "<PJ4/K1ABC>", "DX"]) for _ in range(rng.randrange(1, 5)))
        snr, dt, f = rng.randrange(-30, 10), rng.randrange(-20, 30) / 10, rng.randrange(200, 3000)
        l65 = f"0001 {snr:>3} {dt:>4.1f} {f:>4} #  {msg}"
        lq = f"123045 {snr:>3} {dt:>4.1f} {f:>4} ~  {msg}"
        assert P.parse_decode_line("JT65", l65, 7076000) == oracle.parse_decode_line("JT65", l65, 7076000), l65
        assert P.parse_decode_line("Q65-30", lq, 7076000) == oracle.parse_decode_line("Q65-30", lq, 7076000), lq
