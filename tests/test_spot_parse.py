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
        P.parse_decode_line("WSPR", good, 0)
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
