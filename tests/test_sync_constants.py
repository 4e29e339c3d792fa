"""The sync transform's small twiddles are LITERALS of the kernels (sync_kernels.hpp: small_wr / small_wi, w5_r / w5_i, W3_S; the library also
checks them against its host's libm at start-up and refuses to run on a mismatch).  Here: the literals in the header are float32(cos / -sin) of
the double angle, as the oracle builds its own tables, and the prime-factor index maps of spec v3 (oracle/sync_oracle.c: dft15_pfa8) are what
the header's comments and constants say."""
import math, os, re
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = open(os.path.join(ROOT, "cwsl_digi_amd", "csrc", "sync_kernels.hpp")).read()
PI = 3.14159265358979323846


def _w(k, n):
    return float(np.float32(math.cos(2.0 * PI * k / n))), float(np.float32(-math.sin(2.0 * PI * k / n)))


def _body(name):
    m = re.search(re.escape(name) + r"\(int k\)\s*\{(.*?)\}", HDR, re.S)
    assert m, name
    return [float.fromhex(x.rstrip("f")) for x in re.findall(r"-?0x[0-9a-f.]+p[-+]?\d+f", m.group(1))]


def test_w9_literals():
    for na, nk in ((9, 4),):
        re_vals = _body("small_wr<%d>" % na)
        im_vals = _body("small_wi<%d>" % na)
        assert len(re_vals) == nk and len(im_vals) == nk                      # k = 1 .. NA / 2 (k = 0 is the exact (1, 0))
        for k in range(1, nk + 1):
            assert (re_vals[k - 1], im_vals[k - 1]) == _w(k, na), (na, k)


def test_w5_and_sin_2pi_3_literals():
    r, i = _body("w5_r"), _body("w5_i")
    assert (r[0], i[0]) == _w(1, 5) and (r[1], i[1]) == _w(2, 5)
    m = re.search(r"constexpr float W3_S = (-?0x[0-9a-f.]+p[-+]?\d+)f;", HDR)
    assert float.fromhex(m.group(1)) == float(np.float32(-math.sin(2.0 * PI / 3.0)))


def test_prime_factor_maps_of_spec_v3():
    # a = (5 n1 + 3 n2) mod 15, c = (10 k1 + 6 k2) mod 15  =>  W15^(a c) = W3^(n1 k1) W5^(n2 k2)
    for n1 in range(3):
        for n2 in range(5):
            a = (5 * n1 + 3 * n2) % 15
            for k1 in range(3):
                for k2 in range(5):
                    c = (10 * k1 + 6 * k2) % 15
                    assert (a * c) % 15 == (5 * ((n1 * k1) % 3) + 3 * ((n2 * k2) % 5)) % 15
    live = {n1: sorted((n2, (5 * n1 + 3 * n2) % 15) for n2 in range(5) if (5 * n1 + 3 * n2) % 15 < 8) for n1 in range(3)}
    assert live == {0: [(0, 0), (1, 3), (2, 6)], 1: [(0, 5), (4, 2)], 2: [(2, 1), (3, 4), (4, 7)]}      # the tables of dft15_pfa8 / stage1_pfa15
    # the lanes' output sets: HALF 0 -> k2 = 0, 1, 4; HALF 1 -> k2 = 2, 3: together every c once
    cs = sorted((10 * k1 + 6 * k2) % 15 for k2 in (0, 1, 4, 2, 3) for k1 in range(3))
    assert cs == list(range(15))


def test_upper_bin_packing_of_the_item_loop():
    # wave 0's lanes 33..63 take the bins 961..991: K = 960 + d, d = 15 a + rem; rem <= 7: (row rem, column a) pairs v1 with u2, else (row 15 - rem, column 63 - a) pairs v2 with u1
    seen = set()
    for q in range(33, 64):
        K = 928 + q
        d = K - 960
        a, rem = divmod(d, 15)
        if rem <= 7:
            r, col = rem, a
            assert 15 * (col + 64) + r == K
        else:
            r, col = 15 - rem, 63 - a
            assert 15 * (127 - col) + 15 - r == K
        assert 0 <= r <= 7 and 0 <= col <= 63 and not (r == 0 and col == 0)
        seen.add(K)
    assert seen == set(range(961, 992))
