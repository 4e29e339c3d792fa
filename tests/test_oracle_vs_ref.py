"""CPU: oracle restatement vs the compiled reference headers (oracle/_ref), where that build exists
(the build container; on the GPU box the prebuilt .so travels with the snapshot)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref/libcwsl_ref.so not built (needs /root/reference)")
    return oracle


@pytest.mark.parametrize("fs,f,usb", [(192000, -26000, True), (192000, 90000, True), (192000, 12000, False),
                                      (96000, -30000, True), (48000, 9000, True), (48000, -3000, False)])
def test_bit_identical_audio_and_phasor(ref, fs, f, usb):
    D = fs // 12000
    n = 4 * D * 3000
    iq = ref.synth_iq(fs ^ (f & 0xFFFF), n, fs, tones_hz=[f + 1000.0, f - 1800.0 if not usb else f + 2222.0], amp=1e4)
    a, ta = ref.Demod(fs, float(f), usb=usb).run(iq, trace=True)
    b, tb = ref.RefDemod(fs, float(f), usb=usb).run(iq, trace=True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(ta.view(np.uint64), tb.view(np.uint64))


def test_getters_and_errors(ref):
    r = ref.RefDemod(192000, 0.0)
    R = ref.ref()
    assert (R.ref_ssbd_in_size(r.h), R.ref_ssbd_out_size(r.h), R.ref_ssbd_out_rate(r.h), R.ref_ssbd_delay(r.h)) == (64, 4, 12000, 8)
    for args, msg in (((192000, 97000.0), "low"), ((192000, 93000.0), "high")):
        with pytest.raises(ValueError) as e1:
            ref.RefDemod(*args)
        with pytest.raises(ValueError) as e2:
            ref.Demod(*args)
        assert msg in str(e1.value) and str(e1.value) == str(e2.value)
    with pytest.raises(ValueError) as e1:
        ref.RefDemod(44100, 0.0)
    with pytest.raises(ValueError) as e2:
        ref.Demod(44100, 0.0)
    assert str(e1.value) == str(e2.value) == "Fs/B must be an even integer >= 4"


def test_lowpass_design_matches(ref):
    for order, bw in ((512, 6000 / 192000.), (256, 6000 / 96000.), (128, 0.125)):
        a = np.empty(order, np.float32); b = np.empty(order, np.float32)
        ref.lib().orc_lowpass_design(order, bw, a)
        ref.ref().ref_build_lowpass(order, bw, b)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
