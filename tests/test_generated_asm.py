"""The hand-scheduled assembly streams of the Costas search (sync2d_asm.inc) and of the exact-mode FIR (exact3_asm.inc) are GENERATED files:
what is committed must be what their generators print, and the streams must keep the properties the kernels rely on."""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen(script):
    return subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)], check=True, capture_output=True, text=True).stdout


def _lines(text, macro):
    body = text.split("#define %s \\\n" % macro, 1)[1].split("\n\n", 1)[0]
    return re.findall(r'"([^"\\]*)\\n\\t"', body)


def test_committed_files_are_the_generators_output():
    for script, inc in (("gen_sync2d_asm.py", "sync2d_asm.inc"), ("gen_exact3_asm.py", "lab/exact3_asm.inc"), ("gen_exact4_asm.py", "lab/exact4_asm.inc")):
        assert _gen(script) == open(os.path.join(ROOT, "cwsl_digi_amd", "csrc", inc)).read(), inc


def test_sync2d_stream_reads_and_waits():
    text = _gen("gen_sync2d_asm.py")
    for macro, n_reads, n_adds, n_writes in (("SYNC2D_ASM_SEARCH_NEXT", 63, 36 + 18, 3), ("SYNC2D_ASM_SEARCH_LAST", 42, 36, 0), ("SYNC2D_ASM_C0_ONLY", 21, 18, 3)):
        ls = _lines(text, macro)
        assert sum(l.startswith("ds_read_b64") for l in ls) == n_reads, macro
        assert sum(l.startswith("v_pk_add_f32") for l in ls) == n_adds, macro
        assert sum(l.startswith("ds_write_b64") for l in ls) == n_writes, macro
        assert ls[-1] == "s_waitcnt lgkmcnt(0)"
        # in-order model of the LDS queue: every register an add consumes has been waited for, and never more than 15 reads are outstanding
        issued, done, where = 0, 0, {}
        for l in ls:
            if l.startswith("ds_read_b64"):
                issued += 1
                where[l.split()[1].rstrip(",")] = issued
                assert issued - done <= 15
            elif l.startswith("s_waitcnt lgkmcnt("):
                done = max(done, issued - int(l[len("s_waitcnt lgkmcnt("):-1]))
            elif l.startswith("v_pk_add_f32"):
                ops = [o.strip() for o in l[len("v_pk_add_f32"):].split(",")]
                for o in ops[1:]:
                    assert where.get(o, 0) <= done, (macro, l)


import pytest


@pytest.mark.parametrize("D", [16, 8, 4])
def test_exact3_stream_shape(D):
    ls = _lines(_gen("gen_exact3_asm.py"), "EXACT3_FIR%d_ASM" % D)
    assert sum(l.startswith("ds_read_b128") for l in ls) == 33 * D // 2 and sum(l.startswith("ds_read_b64") for l in ls) == 33
    assert sum(l.startswith("s_load_dwordx") for l in ls) == (66 if D == 16 else 33) and sum(l.startswith("s_waitcnt lgkmcnt(0)") for l in ls) == 33
    muls = [l for l in ls if l.startswith("v_pk_mul_f32")]
    adds = [l for l in ls if l.startswith("v_pk_add_f32")]
    assert len(muls) == 33 * 2 * D + 33 * 2           # D taps x (Re, Im) per step + two per tail
    assert len(adds) == 33 * 2 * (D - 1) + 33 * 2     # D - 1 accumulations x (Re, Im) per step (the first product starts the sum) + two per tail
    # every tap pair and every sample pair of a step is used exactly once per component, in order
    for n in (0, 1, 32):
        pass
    # un-fused arithmetic only
    assert not any("fma" in l or "fmac" in l for l in ls)
    # a product is consumed no sooner than three instructions after it was made (dependent packed operations wait ~11 cycles)
    last_write = {}
    for i, l in enumerate(ls):
        if l.startswith(("v_pk_mul_f32", "v_pk_add_f32")):
            ops = [o.strip().split(" ")[0] for o in l.split(" ", 1)[1].split(",")]
            for o in ops[1:]:
                if o in last_write and o.startswith("v["):
                    assert i - last_write[o] >= 3 or "s_nop" in "".join(ls[last_write[o]:i]), (i, l)
            last_write[ops[0]] = i
