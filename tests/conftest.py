import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand; liboracle.so is plain C."""
    from oracle import oracle as O
    O.build(ref=True)
    return O


# The arithmetic mode of the context under test: assert_frames_match / assert_int16_match demand IDENTICAL BITS while it is exact.
_STATE = {"exact": False}


@pytest.fixture(params=["exact", "fast"])
def ctx(request):
    """A fresh libcwslgpu context on cuda:0, once per arithmetic mode: "exact" is what cwslg_create() gives (the reference's operation
    order, frames bit-identical to the reference chain's), "fast" is cwslg_set_exact(ctx, 0) (fused polyphase form, float audio
    within 1e-5 of frame peak).  Every parity test therefore also runs in the mode whose error must be zero, and the helpers below
    then compare bits.  No fallback: raises without a gfx950 device."""
    import cwsl_digi_amd as P
    c = P.Context(0)
    plain_set_exact = c.set_exact

    def set_exact(on=True):
        plain_set_exact(on)
        _STATE["exact"] = bool(on)
        c.mode = "exact" if on else "fast"
    c.set_exact = set_exact
    c.set_exact(request.param == "exact")
    yield c
    _STATE["exact"] = False
    c.close()


def assert_frames_match(gpu_f32, ref_f32, tol=1e-5):
    """north_star tolerance: max|gpu-ref| <= 1e-5 * max|ref| per frame (hard part 2); in exact mode: the
    reference's bits."""
    if _STATE["exact"]:
        a, b = np.asarray(gpu_f32, np.float32), np.asarray(ref_f32, np.float32)
        assert len(a) == len(b), f"exact mode: frame lengths differ ({len(a)} vs {len(b)})"
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "exact mode: float frame differs from the reference's bits"
        return 0.0
    peak = float(np.abs(ref_f32).max())
    err = float(np.abs(gpu_f32.astype(np.float64) - ref_f32.astype(np.float64)).max())
    assert err <= tol * max(peak, 1e-30), f"max err {err:.3e} vs peak {peak:.3e} (rel {err / max(peak, 1e-30):.3e})"
    return err / max(peak, 1e-30)


def assert_int16_match(gpu_i16, ref_i16, ref_scaled_f32, peak_rel_tol=1e-5):
    """int16 frames must be identical except where the pre-rounding value lies within
    1e-5*peak of a rounding boundary (documented +-1 LSB ties); in exact mode: identical."""
    diff = gpu_i16.astype(np.int32) - ref_i16.astype(np.int32)
    bad = np.nonzero(diff)[0]
    if bad.size == 0:
        return 0
    assert not _STATE["exact"], f"exact mode: {bad.size} int16 samples differ from the reference's"
    assert np.abs(diff[bad]).max() <= 1, "int16 differs by more than 1 LSB"
    peak = float(np.abs(ref_scaled_f32).max())
    x = ref_scaled_f32[bad].astype(np.float64) + 0.5
    dist = np.abs(x - np.round(x))        # distance of (x+0.5) to the nearest integer = truncation boundary
    assert (dist <= peak_rel_tol * peak).all(), "int16 mismatch away from a rounding tie"
    return int(bad.size)
