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
# -DCWSLG_LAB=1: the measured alternatives of every kernel and the environment switches that select them (scripts/, A/B tests).
# The product library above carries one kernel per job and reads no such switch.
LAB_LIB = os.path.join(LIBDIR, "libcwslgpu_lab.so")
BINDIR = os.path.join(HERE, "bin")
# (CWSLG_SKIMMER_BIN / CWSLG_REALTIME_BIN: other builds of the two host programs for the tests that drive them -- the ThreadSanitizer builds of
# scripts/gpu_r5_tsan.sh; test tooling only, the programs themselves read no such variable)
SKIMMER = os.environ.get("CWSLG_SKIMMER_BIN") or os.path.join(BINDIR, "cwsl_gpu_skimmer")
REALTIME = os.environ.get("CWSLG_REALTIME_BIN") or os.path.join(BINDIR, "cwsl_gpu_realtime")

# -ffp-contract=off: every fused multiply-add in the kernels is an explicit __builtin_fmaf and every
# bit-exact sequence (the float32 phasor recurrence, prepareAudio, the synthetic source) is plain * and +.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC",
               "-Wno-unused-value", "-Wno-unused-result",
               # gfx950 issues v_pk_fma_f32 no faster than two v_fma_f32, and SLP-packing the FIR makes hipcc
               # re-read odd-aligned operand pairs from LDS: keep the FMAs scalar
               "-fno-slp-vectorize"]


# CWSLG_HIPCC_EXTRA: extra compiler flags for a measurement build (-D switches of the A/B scripts); part of the source hash, so the library is rebuilt
import shlex
HIPCC_FLAGS += shlex.split(os.environ.get("CWSLG_HIPCC_EXTRA", ""))


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]


HASHFILE = LIB + ".srchash"


def _source_hash():
    """sha256 over the contents of everything the library is built from (file mtimes do not survive a copy of the tree)."""
    import hashlib
    h = hashlib.sha256()
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if os.path.isfile(os.path.join(CSRC, f))]
    deps += [os.path.join(CSRC, sub, f) for sub in ("host", "lab") for f in sorted(os.listdir(os.path.join(CSRC, sub)))]
    deps.append(os.path.join(HERE, "..", "include", "cwsl_gpu.h"))
    for d in deps:
        h.update(os.path.basename(d).encode())
        h.update(open(d, "rb").read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def _stale():
    if not os.path.isfile(LIB) or not os.path.isfile(LAB_LIB) or not os.path.isfile(SKIMMER) or not os.path.isfile(REALTIME) or not os.path.isfile(HASHFILE):
        return True
    return open(HASHFILE).read().strip() != _source_hash()


def build(force=False, verbose=False):
    """Compile csrc/*.hip -> lib/libcwslgpu.so.  Raises if hipcc is missing or compilation fails."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if os.path.isfile(LIB) and not force:      # a deployed tree without the toolchain: the library it came with is the product
            return LIB
        raise RuntimeError("hipcc not found: libcwslgpu.so cannot be built (and there is no CPU fallback)")
    os.makedirs(LIBDIR, exist_ok=True)
    # one builder at a time (N ranks may import together); the library appears atomically
    import fcntl
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():          # another process built it while this one waited
            return LIB
        procs = []
        for lib, extra in ((LIB, []), (LAB_LIB, ["-DCWSLG_LAB=1"])):
            tmp = lib + ".tmp.%d" % os.getpid()
            cmd = [hipcc] + HIPCC_FLAGS + extra + ["-o", tmp] + sources() + ["-ldl"]
            if verbose:
                print(" ".join(cmd))
            procs.append((subprocess.Popen(cmd), tmp, lib, cmd))
        try:
            for pr, tmp, lib, cmd in procs:
                if pr.wait() != 0:
                    raise subprocess.CalledProcessError(pr.returncode, cmd)
            for pr, tmp, lib, cmd in procs:
                os.replace(tmp, lib)
            with open(HASHFILE, "w") as fh:
                fh.write(_source_hash() + "\n")
        finally:
            for pr, tmp, lib, cmd in procs:
                if pr.poll() is None:
                    pr.kill()
                if os.path.exists(tmp):
                    os.remove(tmp)
        build_skimmer(force=True, verbose=verbose)
    return LIB


def build_skimmer(force=False, verbose=False):
    """Compile the Linux host programs (plain C++17 over the C ABI): csrc/host/skimmer_main.cpp -> bin/cwsl_gpu_skimmer and
    csrc/host/realtime_main.cpp -> bin/cwsl_gpu_realtime (the wall-clock-paced ingest harness)."""
    if os.environ.get("CWSLG_SKIMMER_BIN") or os.environ.get("CWSLG_REALTIME_BIN"):
        return SKIMMER
    newest = max(os.path.getmtime(os.path.join(CSRC, "host", f)) for f in os.listdir(os.path.join(CSRC, "host")))
    if not force and all(os.path.isfile(b) and os.path.getmtime(b) >= newest for b in (SKIMMER, REALTIME)):
        return SKIMMER
    os.makedirs(BINDIR, exist_ok=True)
    for name, out in (("skimmer_main.cpp", SKIMMER), ("realtime_main.cpp", REALTIME)):
        cmd = ["g++", "-std=c++17", "-O2", "-Wall", os.path.join(CSRC, "host", name), "-o", out, "-L" + LIBDIR, "-lcwslgpu",
               "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return SKIMMER


if __name__ == "__main__":
    print(build(force=True, verbose=True))
