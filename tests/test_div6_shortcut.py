"""The Costas search divides by 6 in three operations (sync_kernels.hpp: div6_exact) instead of the IEEE division sequence.  The claim
"identical to x / 6.0f for every input that takes the shortcut" is checked here on the CPU: a sample of 2^25 floats spread over all
bit patterns by default, every float with CWSL_TEST_DIV6_ALL=1 (45 s)."""
import os, shutil, subprocess, sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_div6_shortcut_is_the_division(tmp_path):
    exe = str(tmp_path / "div6_check")
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-o", exe, os.path.join(HERE, "div6_check.c"), "-lm"], check=True)
    stride = "1" if os.environ.get("CWSL_TEST_DIV6_ALL") == "1" else "127"
    out = subprocess.run([exe, stride, "0"], check=True, capture_output=True, text=True).stdout.split()
    n, bad_in, bad_out = (int(v) for v in out)
    assert n >= (1 << 32) // int(stride)
    assert bad_in == 0
    assert bad_out > 0          # the guard is not vacuous: denormal-range inputs do differ and must take the division
