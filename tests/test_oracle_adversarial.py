"""CPU: the oracle restatement against the adversarial fixtures made from the compiled reference headers (tests/gen_golden.py,
tests/adversarial.py): empty band beside strong out-of-band carriers, carriers at the band edge, full-scale integer IQ, all-zero IQ,
a peak inside the start-up transient.  Bit-exact."""
import glob
import os

import numpy as np
import pytest

import adversarial as A

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ADV = sorted(glob.glob(os.path.join(GOLD, "adv_*.npz")))


def test_inventory():
    assert sorted(os.path.basename(p)[4:-4] for p in ADV) == sorted(A.CASES)


@pytest.mark.parametrize("path", ADV, ids=os.path.basename)
def test_adversarial_fixture(oracle, path):
    g = np.load(path)
    name = str(g["name"])
    iq = A.make_iq(name)
    if A.iq_crc(iq) != int(g["iq_crc32"]):
        pytest.skip("this numpy/libm builds a different float32 input than the fixture's generator did")
    c = oracle.Channel("FT8", A.FS, A.BLK, A.F)
    assert c.boundary(100) is None                     # discarded first frame: the demodulator keeps running
    assert c.push_many(iq) == len(iq) // A.BLK
    r = c.boundary(115, want_f32=True)
    nv = int(g["n_valid"])
    audio = r["f32"][:nv]
    assert oracle.crc32(audio.view(np.uint32)) == int(g["audio_crc32"])
    assert np.array_equal(audio[:2048].view(np.uint32), g["audio_head_bits"])
    assert np.array_equal(audio[::97].view(np.uint32), g["audio_every_bits"])
    assert np.array([r["factor"]], np.float32).view(np.uint32)[0] == g["factor_bits"][0]
    assert oracle.crc32(r["i16"]) == int(g["i16_crc32"])
    assert int(np.abs(audio).argmax()) == int(g["argmax"])
    if name == "zeros":
        assert not audio.any() and not r["i16"].any() and abs(float(r["factor"]) - 32767.0 * 0.9) < 1e-2   # 32767 / (0 + 1) * 0.90
    if name == "early_peak":
        assert int(g["argmax"]) < 31                   # inside the window's first 31 outputs
