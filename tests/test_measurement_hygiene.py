"""CPU: the committed measurement records are consistent with the evidence they cite, and the product kernels stay clean.

1. profiles/traffic_per_launch.json is what bench.py REPLAYS as roofline.traffic (PMC counters cannot be read from inside the process).
   Every entry names the PMC summary it came from; this test recomputes the bytes from that summary -- FETCH_SIZE (KiB) x 2 + WRITE_SIZE
   (KiB), as MI355X_MICROARCH.md's HBM section prescribes for gfx950 -- and fails on any mismatch.
2. The product library's kernels, compiled here for gfx950: no scratch (private segment) and no flat_ memory instruction in ANY kernel."""
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _summary(path):
    """{kernel base name + first template argument: {counter: value}} of a profiles/*pmc_summary*.txt file."""
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            name = line.strip().replace("void ", "").replace("cwslg::", "").replace(" ", "")
            cur = out.setdefault(name, {})
        elif cur is not None and line.strip():
            k, v = line.split()
            cur[k] = float(v)
    return out


def _find(summary, kernel):
    want = kernel.replace(" ", "")
    base, args = want.split("<")[0], want.split("<")[1].rstrip(">").split(",") if "<" in want else []
    hits = [k for k in summary if k.split("<")[0] == base and ("<" not in k or k.split("<")[1].rstrip(">").split(",")[:len(args[:2])] == args[:2])]
    assert len(hits) == 1, (kernel, sorted(summary))
    return summary[hits[0]]


def test_replayed_traffic_matches_the_pmc_summaries_it_cites():
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_per_launch.json")))
    checked = 0
    for run in tj["runs"]:
        src = run["source"].split()[0]
        path = os.path.join(ROOT, src)
        assert os.path.isfile(path), f"{src} cited by traffic_per_launch.json is not in the tree"
        c = _find(_summary(path), run["kernel"])
        assert c["FETCH_SIZE"] == pytest.approx(run["fetch_size_kb_raw"], rel=2e-4), run
        assert c["WRITE_SIZE"] == pytest.approx(run["write_size_kb_raw"], rel=2e-4), run
        assert (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 == pytest.approx(run["hbm_bytes_per_launch"], rel=2e-4), run
        # and the traffic is not implausible: between the algorithmic bytes and 1.2x of them
        assert run["algorithmic_bytes_per_launch"] <= run["hbm_bytes_per_launch"] <= 1.2 * run["algorithmic_bytes_per_launch"], run
        checked += 1
    for run in tj.get("sync_runs", []):
        path = os.path.join(ROOT, run["source"].split()[0])
        s = _summary(path)
        total = 0.0
        for kname in run["fetch_size_kb_raw"]:
            c = _find(s, kname)
            assert c["FETCH_SIZE"] == pytest.approx(run["fetch_size_kb_raw"][kname], rel=2e-4), (kname, run)
            assert c["WRITE_SIZE"] == pytest.approx(run["write_size_kb_raw"][kname], rel=2e-4), (kname, run)
            total += (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        assert total == pytest.approx(run["fabric_bytes_per_boundary"], rel=2e-4), run
        checked += 1
    assert checked >= 4


@pytest.fixture(scope="module")
def product_isa(tmp_path_factory):
    from cwsl_digi_amd import build as B
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.fail("hipcc is needed to inspect the product kernels")
    # compiled from a copy of the tree WITHOUT csrc/lab/: the product translation unit must not need the retired kernels or their dispatch
    import shutil
    tmp = tmp_path_factory.mktemp("isa")
    shutil.copytree(os.path.join(ROOT, "cwsl_digi_amd", "csrc"), tmp / "cwsl_digi_amd" / "csrc", ignore=shutil.ignore_patterns("lab"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp / "include")
    out = tmp / "prod.s"
    flags = [f for f in B.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    srcs = [str(tmp / "cwsl_digi_amd" / "csrc" / os.path.basename(f)) for f in B.sources()]
    subprocess.check_call([hipcc] + flags + ["-S", "--cuda-device-only", "-o", str(out)] + srcs, stderr=subprocess.DEVNULL)
    s = open(out).read()
    kernels = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        name, blk = m.group(1), m.group(2)
        i = s.find("\n" + name + ":")
        body = s[i:s.find("s_endpgm", i)]
        kernels[name] = dict(scratch=int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", blk).group(1)),
                             scratch_ops=len(re.findall(r"\bscratch_", body)), flat=len(re.findall(r"\bflat_(?:load|store|atomic)", body)),
                             vgpr=int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", blk).group(1)),
                             lds=int(re.search(r"\.amdhsa_group_segment_fixed_size\s+(\d+)", blk).group(1)))
    return kernels


def test_no_product_kernel_uses_scratch(product_isa):
    bad = {k: v for k, v in product_isa.items() if v["scratch"] or v["scratch_ops"]}
    assert not bad, bad


def test_no_product_kernel_has_flat_memory_instructions(product_isa):
    """Every global access of every product kernel goes through the global address space (global_load / global_store): a flat_ access also
    counts in lgkmcnt and stalls the next LDS wait for an HBM round trip.  Round 4 converted the last 139 of them (descriptor pointers of
    the FT8 search, the FT4 and 120 s kernels, the phasor kernels)."""
    bad = {k: v["flat"] for k, v in product_isa.items() if v["flat"]}
    assert not bad, bad
    assert len(product_isa) >= 30


def test_exact_kernel_register_budget(product_isa):
    """Exact mode, demod_exact5_kernel<16 / 8 / 4>: two waves per SIMD = at most 256 VGPRs (the generated statement fixes 241 of them; the rest are its
    operands), 64 KB of LDS per four-wave workgroup (4 waves x 4 row buffers x 4 KB: exactly the static limit, two workgroups per CU); no other
    exact-mode kernel in the product."""
    k = [v for n, v in product_isa.items() if "demod_exact5_kernel" in n]                # 192 / 96 / 48 kHz
    assert len(k) == 3 and all(v["vgpr"] <= 256 for v in k), k
    assert all(v["lds"] == 65536 for v in k), k
    assert not [n for n in product_isa if "demod_exact3" in n or "demod_exact4" in n]
