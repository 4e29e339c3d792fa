"""Build libcwslgpu.so (HIP, gfx950 only) in-tree with hipcc.

The shared library is the product: a C-ABI (include/cwsl_gpu.h) over hand-written gfx950 kernels.
There is no CPU fallback; this module only compiles, it never substitutes another implementation.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcwslgpu.so")

# -ffp-contract=off: every fused multiply-add in the kernels is an explicit __builtin_fmaf and every
# bit-exact sequence (the float32 phasor recurrence, prepareAudio, the synthetic source) is plain * and +.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC",
               "-Wno-unused-value", "-Wno-unused-result",
               # gfx950 issues v_pk_fma_f32 no faster than two v_fma_f32, and SLP-packing the FIR makes hipcc
               # re-read odd-aligned operand pairs from LDS: keep the FMAs scalar
               "-fno-slp-vectorize"]


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]


def _stale():
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(HERE, "..", "include", "cwsl_gpu.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile csrc/*.hip -> lib/libcwslgpu.so.  Raises if hipcc is missing or compilation fails."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libcwslgpu.so cannot be built (and there is no CPU fallback)")
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", LIB] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
