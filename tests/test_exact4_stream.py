"""CPU: the generated FIR streams of the exact mode, run instruction by instruction on a one-lane emulator (tests/asm_emulator.py) against
a numpy float32 restatement of ProcessBlock (SSBD.hpp:160-183): for random mixed samples, phases and taps the streams leave in W exactly
the bits of (Re of output o's workspace slot, Im of output o + 1's).  exact3's stream (one wave, 33 steps) validates the emulator -- that
stream is bit-identical to the compiled reference on the GPU -- and exact4's two streams (steps 0-16 | 17-32 with the hand-over of W_A
through LDS) must give the same bits."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from asm_emulator import Lane

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32
D = 16


def _gen(script):
    return subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)], check=True, capture_output=True, text=True).stdout


def _lines(text, macro):
    body = text.split("#define %s \\\n" % macro, 1)[1].split("\n\n", 1)[0]
    return re.findall(r'"([^"\\]*)\\n\\t"', body)


def _problem(seed, row_bytes, D=D):
    """A lane's window: 33 blocks of D mixed samples + block phases in the two parity arrays, the interleaved tap table, and the reference bits."""
    rng = np.random.default_rng(seed)
    t = (rng.standard_normal((33, D, 2)) * 1000).astype(F)            # t[n][m] = in[m] * tone[m] of block o + n
    ph = np.stack([np.cos(rng.uniform(0, 6.28, 33)), np.sin(rng.uniform(0, 6.28, 33))], 1).astype(F)
    h = (rng.standard_normal(32 * D) * 0.03).astype(F)                # taps h[m + D n]
    # LDS image: block o + n = row n >> 1 of parity array n & 1 (this lane's row 0); row = D samples, then the phase
    nrow = 17
    lds = np.zeros((2, nrow, row_bytes // 4), F)
    for n in range(33):
        lds[n & 1, n >> 1, :2 * D] = t[n].reshape(-1)
        lds[n & 1, n >> 1, 2 * D:2 * D + 2] = ph[n]
    # taps2[n][m] = (h[m + D n], h[m + D (n - 1)]), zero where the tap block does not exist
    taps2 = np.zeros((33, D, 2), F)
    for n in range(33):
        for m in range(D):
            taps2[n, m, 0] = h[m + D * n] if n < 32 else 0
            taps2[n, m, 1] = h[m + D * (n - 1)] if n >= 1 else 0
    # reference: output o uses blocks o .. o + 31 with tap blocks 0..31; output o + 1 uses blocks o + 1 .. o + 32
    def slot(first_block):
        wr, wi = F(0), F(0)
        for k in range(32):
            n = first_block + k
            sr, si = F(0), F(0)
            for m in range(D):
                sr = F(sr + F(t[n, m, 0] * h[m + D * k]))             # sum += (in * tone) * h   (:167-168), un-fused
                si = F(si + F(t[n, m, 1] * h[m + D * k]))
            pr = F(F(sr * ph[n, 0]) - F(si * ph[n, 1]))               # sum * phase (:170)
            pi = F(F(sr * ph[n, 1]) + F(si * ph[n, 0]))
            wr, wi = F(wr + pr), F(wi + pi)
        return wr, wi
    want = np.array([slot(0)[0], slot(1)[1]], F)
    return lds, taps2, want


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("d", [16, 8, 4])
def test_exact3_stream_on_the_emulator_validates_it(seed, d):
    """All three decimations' streams (192 / 96 / 48 kHz; the 96 and 48 kHz ones are the product's exact kernels at those rates)."""
    text = _gen("gen_exact3_asm.py")
    row_bytes = (d + 2) * 8
    lds, taps2, want = _problem(seed, row_bytes, d)
    lane = Lane(lds.tobytes(), taps2.tobytes(), r0=0, r1=lds[0].nbytes)
    got = lane.run(_lines(text, "EXACT3_FIR%d_ASM" % d))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (got, want)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_exact4_streams_hand_over_and_match_the_reference_bits(seed):
    text = _gen("gen_exact4_asm.py")
    assert text == open(os.path.join(ROOT, "cwsl_digi_amd", "csrc", "lab", "exact4_asm.inc")).read(), "exact4_asm.inc is not the generator's output"
    row_bytes = int(re.search(r"#define EXACT4_ASM_ROW_BYTES (\d+)", text).group(1))
    split = int(re.search(r"#define EXACT4_ASM_SPLIT (\d+)", text).group(1))
    lds, taps2, want = _problem(seed, row_bytes)
    image = lds.tobytes()
    xa = len(image)                                                   # the exchange slot sits behind the image
    a_lines, b_lines = _lines(text, "EXACT4_FIRA_ASM"), _lines(text, "EXACT4_FIRB_ASM")
    A = Lane(image + bytes(8), taps2.tobytes(), r0=0, r1=lds[0].nbytes, xa=xa)
    w_a = A.run(a_lines)
    B = Lane(image + bytes(8), taps2.tobytes(), r0=0, r1=lds[0].nbytes, xa=xa)
    def publish(lane):                                                # what stream A's wave does before ITS barrier: ds_write_b64 W_A
        lane.lds[xa // 4:xa // 4 + 2] = w_a.view(np.uint32)
    B.on_barrier = publish
    got = B.run(b_lines)
    assert A.barriers == 0 and B.barriers == 1
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (got, want)
    # shape: the two streams together are exact3's operations -- 33 x 2 D products and 33 x 2 (D - 1) accumulations, two products per tail
    both = a_lines + b_lines
    assert sum(l.startswith("v_pk_mul_f32") for l in both) == 33 * 2 * D + 33 * 2
    assert sum(l.startswith("v_pk_add_f32") for l in both) == 33 * 2 * (D - 1) + 33 * 2
    assert sum(l.startswith("s_load_dwordx16") for l in a_lines) == 2 * split and sum(l.startswith("s_load_dwordx16") for l in b_lines) == 2 * (33 - split)
    assert not any("fma" in l for l in both)
    # registers: stream A stays inside v72..v127, stream B inside v40..v127 (four waves per SIMD: 128 registers per lane)
    for lines, lo in ((a_lines, 72), (b_lines, 40)):
        regs = [int(x) for l in lines for x in re.findall(r"v\[(\d+):\d+\]", l)] + [int(x) for l in lines for x in re.findall(r"\bv(\d+)\b", l)]
        assert min(regs) >= lo and max(regs) <= 127
    # a product is consumed no sooner than three instructions after it was made
    for lines in (a_lines, b_lines):
        last_write = {}
        for i, l in enumerate(lines):
            if l.startswith(("v_pk_mul_f32", "v_pk_add_f32")):
                ops = [o.strip().split(" ")[0] for o in l.split(" ", 1)[1].split(",")]
                for o in ops[1:]:
                    if o in last_write and o.startswith("v[") and not l.startswith("v_pk_add_f32 %[w]"):
                        assert i - last_write[o] >= 3 or "s_nop" in "".join(lines[last_write[o]:i]), (i, l)
                last_write[ops[0]] = i
