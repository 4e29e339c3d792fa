"""CPU: the C-ABI library loads and exports every symbol include/cwsl_gpu.h declares (no compute calls
without a GPU), fails loudly without a device, and the host-side tables mirror the reference."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "cwsl_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cwslg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import cwsl_digi_amd as P
    from cwsl_digi_amd import api
    L = P.load_library()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in cwsl_gpu.h but not exported"
    assert sorted(api.ABI_SYMBOLS) == names, set(api.ABI_SYMBOLS) ^ set(names)
    assert L.cwslg_abi_version() == 5


def test_fails_loudly_without_gpu():
    import torch
    import cwsl_digi_amd as P
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(P.CwslGpuError) as e:
        P.Context(0)
    assert e.value.status == -7 and "no CPU fallback" in str(e.value)


def test_strerror_mirrors_reference_messages():
    import cwsl_digi_amd as P
    L = P.load_library()
    assert L.cwslg_strerror(-1) == b"Fs/B must be an even integer >= 4"      # SSBD.hpp:55
    assert L.cwslg_strerror(-2) == b"Signal outside of band (low)"           # SSBD.hpp:101
    assert L.cwslg_strerror(-3) == b"Signal outside of band (high)"          # SSBD.hpp:103


def test_null_context_is_rejected():
    import cwsl_digi_amd as P
    L = P.load_library()
    assert L.cwslg_process(None) == -6
    assert L.cwslg_slot_boundary(None, 0, 0) == -6
    rid = ctypes.c_int()
    assert L.cwslg_receiver_open(None, 192000, 2048, 0, 0, ctypes.byref(rid)) == -6


def test_mode_tables_match_oracle(oracle):
    import cwsl_digi_amd as P
    from cwsl_digi_amd import api
    for m in api._MODE_PERIOD:
        assert P.frame_len(m) == oracle.frame_len(m), m
    # SyncPredicates groups (CWSL_DIGI_Types.hpp:83-134)
    assert P.group_of("FT8") == P.group_of("JS8") == P.GROUPS["FT8"]
    assert P.group_of("WSPR") == P.group_of("FST4-120") == P.group_of("FST4W-120") == P.GROUPS["S120"]
    assert P.group_of("JT65") == P.group_of("FST4-60") == P.GROUPS["S60"]
    assert len({P.group_of(m) for m in api._MODE_PERIOD}) == 8


def test_no_product_import_of_oracle():
    """The product path must never route through the oracle."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cwsl_digi_amd")):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                txt = open(path).read()
                assert not re.search(r"^\s*(import|from)\s+oracle", txt, flags=re.M), f
                assert "liboracle" not in txt and "_ref" not in txt, f
            elif f.endswith((".hip", ".hpp", ".inc", ".h")):
                txt = open(path).read()
                assert not re.search(r"#\s*include\s*[<\"][^>\"]*oracle", txt), f
                # the only library the product may open at run time is RCCL (multi_gpu.inc); never the oracle
                for m in re.finditer(r"dlopen\(([^,]*),", txt):
                    assert m.group(1).strip() == "n", (f, m.group(0))
                if "dlopen(" in txt:
                    assert f == "multi_gpu.inc" and re.findall(r'"([^"]*\.so[^"]*)"', txt) == \
                        ["librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"], f


def test_shim_header_compiles():
    """include/cwsl_gpu_shim.hpp (SSBD-/Receiver-shaped C++ wrappers) is self-contained C++17."""
    import subprocess
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror",
                           os.path.join(ROOT, "tests", "shim_compile_check.cpp")])


def _kernel_names(path):
    """Names of the gfx950 kernels in a shared library's code object (their .kd descriptors appear in its text)."""
    data = open(path, "rb").read()
    names = set(m.decode() for m in re.findall(rb"(_Z[A-Za-z0-9_]+)\.kd", data))
    return names


def test_product_library_has_one_kernel_per_job_and_no_lab_switches():
    """The shipped libcwslgpu.so carries ONE kernel per job and reads no environment switch that changes a kernel or the order of its
    arithmetic; the measured alternatives (round-1/-2 kernels, matrix-core and persistent variants, probes) and the CWSLG_*_VARIANT
    switches live in libcwslgpu_lab.so only."""
    from cwsl_digi_amd import build as B
    B.build()
    prod, lab = open(B.LIB, "rb").read(), open(B.LAB_LIB, "rb").read()
    for switch in (b"CWSLG_DEMOD_VARIANT", b"CWSLG_SYNC_VARIANT", b"CWSLG_LONG_VARIANT", b"CWSLG_FT4_DFT", b"CWSLG_ITEM_ORDER",
                   b"CWSLG_UPLOAD", b"CWSLG_COPY_ON_MAIN", b"CWSLG_PERSIST_WGS_PER_CU", b"CWSLG_EXACT5_SEG", b"CWSLG_EXACT5_SEG_FORCE"):
        assert switch not in prod, switch
        assert switch in lab, switch
    kp, kl = _kernel_names(B.LIB), _kernel_names(B.LAB_LIB)
    assert kp < kl                                      # the lab library has everything the product has, and more
    for lab_only in ("ring_probe_kernel", "demod_mfma1p_kernel", "demod_mfma_bf16_kernel", "demod_exact_kernel", "demod_exact2_kernel", "demod_exact3_kernel", "demod_exact4_kernel",
                     "ft8_sync2d_kernelE", "ft8_sync2d_v2_kernel", "ft8_candidates_kernelILi256E", "symbol_spectra_kernelI",
                     "ft4_dft567_kernelE"):
        assert not any(lab_only in k for k in kp), lab_only
        assert any(lab_only in k for k in kl), lab_only
    # one demod kernel per mode and sample rate (D = 16, 8, 4), nothing persistent or alternative
    demod = sorted(k for k in kp if "demod_kernel" in k)
    # (tile by decimation: 256 outputs at 192 kHz, 512 at 96 kHz, 768 at 48 kHz -- the same amount of IQ per workgroup)
    assert len(demod) == 3 and sum(t in k for k in demod for t in ("ILi16ELi256ELi256ELi0E", "ILi8ELi512ELi256ELi0E", "ILi4ELi768ELi256ELi0E")) == 3, demod
    exact = sorted(k for k in kp if "demod_exact" in k)
    # exact mode: ONE kernel per rate -- the stream form (round 5: lane = stream, K = 1 matrix products), a demodulator's first outputs included; the tile
    # kernels of rounds 3 and 4 are lab-only
    assert exact == sorted("_ZN5cwslg19demod_exact5_kernelILi%dEEEvPKNS_8ChanWorkEPKfiiiPy" % d for d in (16, 8, 4)), exact
    assert any("demod_exact4_kernel" in k for k in kl) and any("demod_exact3_kernel" in k for k in kl)
    # FT8: Costas search + candidate selection in one launch per boundary, or (few channels) one workgroup per band + the selection
    assert any("ft8_sync_chan_kernel" in k for k in kp) and any("ft8_sync2d_v3_kernel" in k for k in kp)
    assert len(kp) <= 37, sorted(kp)                   # round 4: + scatter_blocks_kernel (cwslg_push_iq_many); round 5: exact5<16 / 8 / 4> for exact4 + exact3<8 / 4>


def test_exact_kernel_launch_policy_is_what_the_notebook_says():
    """cwslg_exact_stream_length (pure): outputs per stream of demod_exact5_kernel by launch size.  Every stream pays a 32-block warm-up, so the
    redundancy of a launch is 1 + 32 / length (docs/lab-notebook.md, "Round 6: launch policy").  No GPU needed."""
    import cwsl_digi_amd as P
    L = P.exact_stream_length
    red = lambda n: 1.0 + 32.0 / n
    assert L(4096 * 180000, 180000, 256, latency=True) == 1408 == L(4096 * 180000, 180000, 256, latency=False)      # the bench: capped, 1.023
    assert L(4096 * 128, 128, 256, latency=True) == 4 and red(4) == 9.0                                             # a launch per block: nine-fold
    assert L(4096 * 128, 128, 256, latency=False) == 4                                                               # (nothing to gain: 128 / 32 = 4)
    assert L(1024 * 20480, 20480, 256, latency=False) == 640 and abs(red(640) - 1.05) < 1e-9                        # the library's own threshold: 5 %
    assert L(1024 * 20480, 20480, 256, latency=True) == 160                                                          # the same launch, were somebody waiting
    assert L(4096 * 28800, 28800, 256, latency=False) == 900 == L(4096 * 28800, 28800, 256, latency=True)           # ring pressure at 4096 channels (2.4 s pending)
    assert L(10 * 28800, 28800, 256, latency=True) == 4 and L(10 * 28800, 28800, 256, latency=False) == 900         # ten decoders: short launch at the boundary, one wave each otherwise
    for n in (1, 4, 100, 10 ** 6, 10 ** 10):
        for lat in (True, False):
            v = L(n, min(n, 10 ** 6), 256, lat)
            assert 4 <= v <= 1408 and v % 4 == 0
