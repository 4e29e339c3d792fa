"""GPU: the C++ shim (SSBD-/Receiver-shaped wrappers over the C ABI) built into a real host program and run."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shim_program_end_to_end(ctx, oracle, tmp_path):
    lib = os.path.join(ROOT, "cwsl_digi_amd", "lib")
    exe = str(tmp_path / "shim_run")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "shim_run.cpp"), "-o", exe,
                           "-L" + lib, "-lcwslgpu", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    n = 64 * 2048
    iq = oracle.synth_iq(41, n, 192000, tones_hz=[-26000 + 1300.0], amp=1.5e4)
    p = tmp_path / "iq.c64"
    iq.tofile(p)
    out = subprocess.run([exe, str(p)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    tag, t0, nsamp, crc, in_size, out_rate = out.stdout.split()
    assert (tag, int(t0), int(nsamp), int(in_size), int(out_rate)) == ("OK", 1000, 240000, 64, 12000)
    # same input through the ctypes mirror of the same library, in the mode a new context has (the shim program sets none)
    ctx.set_exact(True)
    rx = ctx.receiver_open(192000, 2048, 28100000)
    ch = ctx.channel_open(rx, -26000, "FT8")
    ctx.slot_boundary("FT8", 1000); ctx.push_iq(rx, iq); ctx.slot_boundary("FT8", 1015)
    assert int(crc, 16) == oracle.crc32(ctx.fetch_frame(ch)["i16"])
