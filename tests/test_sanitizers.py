"""CPU: the host side of the library under AddressSanitizer + UndefinedBehaviorSanitizer ("race detection / sanitizers").
cwsl_gpu.hip is rebuilt with -fsanitize=address,undefined on its HOST code only (-fno-gpu-sanitize: GPU sanitizers are not available on this pool)
and the tests of its pure-function half -- decoder= grammar, slot clock, pool sizing, spot parsing, the decoder hand-off block and commands, WAV
header -- run against that build in a child process with the sanitizer runtime preloaded.  Any report fails the test.
The threaded host programs get a ThreadSanitizer build on the GPU box: scripts/gpu_r6_tsan.sh (profiles/r6_tsan.txt; round 5: gpu_r5_tsan.sh, r5_tsan.txt)."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_functions_are_clean_under_asan_and_ubsan(tmp_path):
    from cwsl_digi_amd import build as B
    hipcc = "/opt/rocm/bin/hipcc"
    rt = glob.glob("/opt/rocm*/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not os.path.exists(hipcc) or not rt:
        pytest.fail("hipcc and its AddressSanitizer runtime are needed for the sanitizer build")
    lib = str(tmp_path / "libcwslgpu_asan.so")
    flags = [f for f in B.HIPCC_FLAGS if f != "-O3"] + ["-O1", "-g", "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-shared-libsan"]
    subprocess.check_call([hipcc] + flags + ["-o", lib] + B.sources() + ["-ldl"], stderr=subprocess.DEVNULL)
    env = dict(os.environ, CWSLG_LIB=lib, LD_PRELOAD=rt[0], ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "not gpu", "-p", "no:cacheprovider", "tests/test_host_service.py", "tests/test_spot_parse.py",
                        "tests/test_decoder_lines.py", "tests/test_handoff.py", "tests/test_abi_and_host.py"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = p.stdout + p.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert p.returncode == 0 and " passed" in out, out[-4000:]
