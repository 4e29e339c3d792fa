"""CPU: config.ini `decoder=` lines are accepted unchanged (north_star) -- the grammar of CWSL_DIGI.cpp:731-837."""
import numpy as np
import pytest

# the example lines shipped in the reference's config.ini (lines 46-145), as data
CONFIG_INI_LINES = """50323000 FT8|50318000 FT4|50313000 FT8|50310000 FT8|50275000 Q65-30|28180000 FT4|28074000 FT8|28076000 JT65|
28078000 JS8|24919000 FT4|24915000 FT8|24917000 JT65|21140000 FT4|21094600 WSPR|21074000 FT8|21076000 JT65|21078000 JS8|
18104600 WSPR|18104000 FT4|18100000 FT8|18102000 JT65|14095600 WSPR|14090000 FT8|14080000 FT4|14074000 FT8|14076000 JT65|
14078000 JS8|10140000 FT4|10138700 WSPR|10136000 FT8|10131000 FT8|10138000 JT65|10130000 JS8|7074000 FT8|7047500 FT4|
7038600 WSPR|7076000 JT65|7078000 JS8|5357000 FT8|5287200 WSPR""".replace("\n", "").split("|")


def test_reference_config_lines_parse():
    import cwsl_digi_amd as P
    for line in CONFIG_INI_LINES:
        d = P.parse_decoder_line(line)
        f, m = line.split(" ")
        assert d["freq_hz"] == int(f) == d["calibrated_hz"] and d["mode"] == m
        assert d["smnum"] == -1 and d["freqcal"] == 1.0 and d["callsign"] == ""
        assert d["frame_len"] == P.frame_len(m) and d["group"] == P.group_of(m)


def test_optional_fields_and_calibration():
    import cwsl_digi_amd as P
    d = P.parse_decoder_line("14074000 FT8 2")
    assert d["smnum"] == 2
    d = P.parse_decoder_line("14074000 FT8 1 1.0000035", freqcal_global=0.9999990)
    # :834  static_cast<FrequencyHz>(freq / (freqCalGlobal * decoder_freqcal))
    assert d["calibrated_hz"] == int(np.uint32(14074000 / (0.9999990 * 1.0000035)))
    assert d["freqcal"] == 1.0000035
    d = P.parse_decoder_line("14095600 WSPR 0 1.0 W1AW")
    assert d["callsign"] == "W1AW" and d["mode"] == "WSPR"
    assert P.parse_decoder_line("1840000 FST4W-1800")["frame_len"] == 12000 * 1805


@pytest.mark.parametrize("line,status", [
    ("14074000", -6),                       # one field
    ("14074000 FT8 1 1.0 W1AW", -6),        # callsign only for WSPR (:825-828)
    ("14074000 WSPR 1 1.0 W1AW extra", -6), # six fields
    ("14074000 PSK31", -5),                 # unknown mode (:797-801)
    ("14074000  FT8", -5),                  # double space -> empty mode field (getline semantics)
    ("abc FT8", -6),                        # std::stoi throws
    ("14074000 FT8 x", -6),
    ("14074000 FT8 1 y", -6),
    ("", -6),
])
def test_error_cases(line, status):
    import cwsl_digi_amd as P
    with pytest.raises(P.CwslGpuError) as e:
        P.parse_decoder_line(line)
    assert e.value.status == status


def test_stoi_like_prefix_parsing():
    import cwsl_digi_amd as P
    assert P.parse_decoder_line("7074000Hz FT8")["freq_hz"] == 7074000     # stoi stops at the first non-digit
    assert P.parse_decoder_line("14074000 FT8 ")["mode"] == "FT8"          # trailing space: no extra field
