"""CPU: the generated program of demod_exact5_kernel (scripts/gen_exact5_asm.py: one wave's whole life -- prologue, tile loop with its branches, epilogue)
run on the 64-lane emulator (tests/wave_emulator.py) against the oracle's demodulator (oracle/cwsl_oracle.c, pinned bit for bit to the compiled
SSBD.hpp): the 32 streams of a wave write exactly the reference's float32 audio, the ragged last lanes write nothing, the ring wrap inside a
stream is followed, the peak is the maximum of what was written.  The emulator also re-derives every wait the text needs (memory / LDS loads in
flight, MFMA results, operands of the matrix instruction and of the lane swap) and fails on an uncovered access."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from wave_emulator import Wave

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, U = np.float32, np.uint32


def _gen(dma=None, extra=None):
    env = dict(os.environ)
    env.update(extra or {})
    if dma is not None:
        env["X5_DMA"] = "1" if dma else "0"
    return subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_exact5_asm.py")], check=True, capture_output=True, text=True, env=env).stdout


def _lines(text, macro):
    body = text.split("#define %s \\\n" % macro, 1)[1].split('    ""\n', 1)[0]
    return re.findall(r'"([^"\\]*)\\n\\t"', body)


def _define(text, name):
    return int(re.search(r"#define %s (\d+)" % name, text).group(1))


def _cmul(a, b):
    return (F(F(a[0] * b[0]) - F(a[1] * b[1])), F(F(a[0] * b[1]) + F(a[1] * b[0])))


def _tap_block_of_row(i):
    return (i & 3) + 4 * (i >> 3) + 16 * ((i >> 2) & 1)


def _run_wave(oracle, f_hz, usb, seg_len, n_blocks, seed, ring_shift_blocks, D=16, q_first=32, dma=None, extra=None):
    text = _gen(dma, extra)
    dma = bool(_define(text, "EXACT5_ASM_DMA"))
    FS = 12000 * D
    row, buf, per_iter = _define(text, "EXACT5_ASM_ROW_BYTES"), _define(text, "EXACT5_ASM_BUF_BYTES"), _define(text, "EXACT5_D%d_TILES_PER_ITER" % D)
    # q_first = block index (since the demodulator's creation) of the wave's first output; below 32, stream 0's warm-up reaches before the origin
    total_blocks = q_first + n_blocks
    n_samp = total_blocks * D
    iq = oracle.synth_iq(seed, n_samp, FS, tones_hz=[f_hz + 700.0, f_hz + 2250.5], amp=1.7e4)
    dm = oracle.Demod(FS, f_hz, usb=usb)
    ref = dm.run(iq)
    taps, tone, inc = dm.taps.astype(F), dm.tone, dm.phase_inc
    # ---- global memory image ----
    cap = ((n_samp + 64 * 37) // 64) * 64                     # ring capacity in samples, a multiple of 64
    # logical sample i lives at ring[(i + shift) % cap]; a negative argument puts the ring's END that many blocks into the data (4-block granularity,
    # the push granularity of the library: 512 / 256 / 128 bytes at 192 / 96 / 48 kHz)
    shift = ring_shift_blocks * D if ring_shift_blocks >= 0 else cap + ring_shift_blocks * D
    ring = (np.arange(2 * cap, dtype=np.float32) * 0 + 7.0e8)  # junk everywhere the stream must not read
    ring = ring.reshape(cap, 2)
    pos = (np.arange(n_samp) + shift) % cap
    ring[pos, 0], ring[pos, 1] = iq.real, iq.imag
    n_ck = total_blocks // 4 + 2
    ck = np.zeros((n_ck, 2), F)
    p = (F(1), F(0))
    for q in range(4 * n_ck):
        if q % 4 == 0:
            ck[q // 4] = p
        p = _cmul(p, (F(inc.real), F(inc.imag)))
    tone_ri = np.stack([tone.real, tone.imag], 1).astype(F)
    out = np.full(n_blocks + 64, 3.0e8, F)
    parts, base = {}, {}
    cur = 256
    for name, arr in (("ring", ring), ("taps", taps), ("tone", tone_ri), ("ckpt", ck), ("out", out)):
        base[name] = cur
        parts[name] = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        cur += (len(parts[name]) + 255) // 256 * 256
    mem = np.zeros(cur, np.uint8)
    for name in parts:
        mem[base[name]:base[name] + len(parts[name])] = parts[name]
    # ---- operands of the statement, as the kernel's C++ prologue computes them ----
    lane = np.arange(64)
    j = lane & 31
    def stream_pos_bytes(s):                                   # ring byte offset of the first sample of stream s (its 32-block warm-up included)
        qs = q_first + s * seg_len - 32
        return (((qs * D + shift) % cap) * 8).astype(np.int64)
    n_out = np.clip(n_blocks - j * seg_len, 0, seg_len)
    ops = {}
    if dma:                                                    # the source-side swizzle: lane l of load i fetches piece (l & 7) ^ f(row) of row 8 i + (l >> 3), f(row) = (row >> 1) & 7
        piece = lambda i: ((lane & 7) ^ (((8 * i + (lane >> 3)) >> 1) & 7)) * 16
        for i in range(4):
            ops["off%d" % i] = ("v", (stream_pos_bytes(8 * i + (lane >> 3)) + piece(i)).astype(U))
        for par in (0, 1):
            ops["capl%d" % par] = ("v", (cap * 8 + piece(par)).astype(U))
            ops["pc16%d" % par] = ("v", piece(par).astype(U))
        ops["ldsr"] = ("v", (j * 128).astype(U))
        ops["ldsw"] = ("v", (lane * 16).astype(U))
        ops["fj16"] = ("v", (((j >> 1) & 7) * 16).astype(U))
    else:
        for i in range(4):
            ops["off%d" % i] = ("v", (stream_pos_bytes(8 * i + (lane >> 3)) + (lane & 7) * 16).astype(U))
        ops["capl"] = ("v", (cap * 8 + (lane & 7) * 16).astype(U))
        ops["pc16"] = ("v", ((lane & 7) * 16).astype(U))
        ops["ldsr"] = ("v", (j * row).astype(U))
        ops["ldsw"] = ("v", ((lane >> 3) * row + (lane & 7) * 16).astype(U))
    ops["ckoff"] = ("v", (np.maximum(q_first + j * seg_len - 32, 0) // 4 * 8).astype(U))
    ops["tapoff"] = ("v", np.array([4 * D * _tap_block_of_row(int(i)) for i in j], U))
    ops["rem"] = ("v", n_out.astype(U))
    ops["outoff"] = ("v", (j * seg_len * 4).astype(U))
    ops["peak"] = ("v", np.zeros(64, U))
    # pre-origin counts: tiles of this lane's stream (low half), load tiles of this LOADER lane's first stream (high half)
    ops["pk"] = ("v", (np.maximum(32 - q_first - j * seg_len, 0) | ((np.maximum(32 - q_first - (lane >> 3) * seg_len, 0) * D // 16) << 16)).astype(U))
    sign = F(1.0 if usb else -1.0)
    for name in ("ring", "taps", "tone", "ckpt", "out"):
        ops[name] = ("s64", base[name])
    if dma:
        for k in range(4):
            ops["ring%d" % k] = ("s64", base["ring"] + 128 * k)
        ops.update(ldsb=("s", 0), m0keep=("s", 0))
    ops.update(incre=("s", int(np.array(inc.real, F).view(U))), incim=("s", int(np.array(inc.imag, F).view(U))),
               sign=("s", int(np.array(sign, F).view(U))), nsign=("s", int(np.array(-sign, F).view(U))),
               hmask=("s64", 0xFFFFFFFF00000000), esave=("s64", 0),
               hold=("s", max(0, 32 - q_first)), holdlt=("s", max(0, 32 - q_first) * D // 16), st1=("s", seg_len * D // 2), st2=("s", seg_len * D), st3=("s", 3 * seg_len * D // 2), warm=("s", _define(text, "EXACT5_ASM_WARM_STORES")), iters=("s", (32 + int(n_out.max()) + per_iter - 1) // per_iter))
    w = Wave(mem, _define(text, "EXACT5_ASM_NBUF") * buf, ops)
    assert w.n_operand_vgprs <= (31 if dma else 15)            # what is left beside the program's fixed registers
    w.run(_lines(text, "EXACT5_D%d_PROLOGUE_ASM" % D) + _lines(text, "EXACT5_D%d_LOOP_ASM" % D) + _lines(text, "EXACT5_D%d_EPILOGUE_ASM" % D))
    got = mem[base["out"]:base["out"] + len(parts["out"])].view(F)
    peak = w.v[w.names["peak"][1]].view(F)
    return got, ref[q_first:], n_out, peak, w, text


@pytest.mark.parametrize("d,f_hz,usb,seed,shift", [(16, -26000, True, 11, 0), (16, 48000, True, 12, 8), (16, 1234, False, 13, -100),
                                                   (8, -26000, True, 14, 0), (8, 30000, False, 15, -100), (8, 5000, True, 18, -140),
                                                   (4, 11000, True, 16, 0), (4, -20500, True, 17, -100), (4, 800, False, 19, -204)])
def test_exact5_wave_program_writes_the_reference_bits(oracle, d, f_hz, usb, seed, shift):
    """All three decimations (192 / 96 / 48 kHz: 16, 8, 4 samples per block; a 128-byte row of the transposing LDS image then holds 1, 2 or 4 tiles)."""
    seg_len, n_blocks = 8, 244                                 # 30 full lanes, one of 4 outputs, one idle; a negative shift puts the ring's end inside a stream
    got, want, n_out, peak, w, text = _run_wave(oracle, f_hz, usb, seg_len, n_blocks, seed, shift, D=d)
    assert np.array_equal(got[:n_blocks].view(U), want[:n_blocks].view(U)), np.nonzero(got[:n_blocks].view(U) != want[:n_blocks].view(U))[0][:8]
    assert (got[n_blocks:] == F(3.0e8)).all(), "a lane wrote past its own outputs"
    assert np.abs(want[:n_blocks]).max() == peak.max() and (peak[:32] == 0).all()
    # shape of a tile: the reference's arithmetic and nothing fused -- 16 matrix instructions (2048 products each), 15 x 32 ordered additions
    loop = _lines(text, "EXACT5_D%d_LOOP_ASM" % d)
    tiles = _define(text, "EXACT5_D%d_TILES_PER_ITER" % d)
    assert tiles * d == 64 and sum(l.startswith("v_mfma_f32_32x32x1_2b_f32") for l in loop) == tiles * d
    # the ordered sums: 15 x 32 additions per tile, issued as register pairs (v_pk_add_f32: two additions, each rounded on its own)
    assert sum(bool(re.match(r"v_pk_add_f32 v\[(\d+):(\d+)\], v\[\1:\2\], v\[(3[2-9]|[4-9]\d):\d+\]$", l)) for l in loop) == tiles * (d - 1) * 16
    assert not any("fma" in l.split(" ")[0].replace("v_mfma", "") for l in loop)
    regs = [int(x) for l in loop for x in re.findall(r"\bv(\d+)\b", l)] + [int(x) for l in loop for x in re.findall(r"v\[\d+:(\d+)\]", l)]
    vtop = int(re.search(r"#define EXACT5_ASM_VTOP (\d+)", text).group(1))              # fixed registers: v0 .. VTOP - 1 and the second staging set v240 .. v255
    assert all(r < vtop or 240 <= r <= 255 for r in regs)


@pytest.mark.parametrize("d,q_first,shift", [(16, 0, 0), (16, 12, -8), (16, 28, 40), (8, 0, 0), (8, 20, -4), (4, 0, 0), (4, 8, -12)])
def test_exact5_first_outputs_of_a_demodulator(oracle, d, q_first, shift):
    """The wave whose first output is one of the demodulator's first 32: stream 0's warm-up blocks precede the origin (x[i < 0] = 0, phase (1, 0) at block 0:
    SSBD.hpp:117-121).  Their rows -- here junk, or the END of the ring's data when the origin sits at the ring's start -- are zeroed on their way into LDS
    and the stream's phase is held; the other 31 streams of the wave run as ever.  Bits of every output, from the very first."""
    seg_len, n_blocks = 8, 200
    got, want, n_out, peak, w, text = _run_wave(oracle, 9000 - 300 * q_first, True, seg_len, n_blocks, 40 + q_first, shift, D=d, q_first=q_first)
    assert np.array_equal(got[:n_blocks].view(U), want[:n_blocks].view(U)), np.nonzero(got[:n_blocks].view(U) != want[:n_blocks].view(U))[0][:8]
    assert (got[n_blocks:] == F(3.0e8)).all()


@pytest.mark.parametrize("d,q_first,shift", [(16, 32, -100), (16, 0, 0), (8, 32, -140), (8, 20, -4), (4, 32, -204), (4, 8, -12)])
def test_exact5_register_staged_form(oracle, d, q_first, shift):
    """The generator's other form of the same program (X5_DMA=0: rows through staging registers and ds_write_b128 into a pitch-144 image, two buffers) --
    the form of the round's first half, kept as the A/B partner of the LDS-DMA form the product runs (profiles/r5_x5_dma_ab.txt)."""
    seg_len, n_blocks = 8, 200
    got, want, n_out, peak, w, text = _run_wave(oracle, 9000 - 100 * q_first, True, seg_len, n_blocks, 60 + q_first + d, shift, D=d, q_first=q_first, dma=False)
    assert np.array_equal(got[:n_blocks].view(U), want[:n_blocks].view(U)), np.nonzero(got[:n_blocks].view(U) != want[:n_blocks].view(U))[0][:8]
    assert (got[n_blocks:] == F(3.0e8)).all()
    assert w.count.get("global_load_lds_dwordx4", 0) == 0 and w.count.get("ds_write_b128", 0) > 0


@pytest.mark.parametrize("d,q_first,shift", [(16, 32, -100), (8, 20, -4), (4, 8, -12)])
def test_exact5_one_lane_sums_form(oracle, d, q_first, shift):
    """X5_PK=0: the block sums as one-lane v_add_f32 (the product's form has them as v_pk_add_f32 on register pairs: the same bits) -- the A/B partner."""
    seg_len, n_blocks = 8, 200
    got, want, n_out, peak, w, text = _run_wave(oracle, 9000 - 100 * q_first, True, seg_len, n_blocks, 90 + d, shift, D=d, q_first=q_first, extra={"X5_PK": "0"})
    assert np.array_equal(got[:n_blocks].view(U), want[:n_blocks].view(U))
    assert w.count.get("v_pk_add_f32", 0) == 0


@pytest.mark.parametrize("d,q_first,shift", [(16, 32, -100), (16, 0, 0), (8, 32, -140), (8, 20, -4), (4, 32, -204), (4, 8, -12)])
def test_exact5_three_load_tiles_in_flight(oracle, d, q_first, shift):
    """X5_DEEP=1: the wait for a load tile's rows just before their first read, three load tiles outstanding (the emulator refuses a read of LDS bytes that an
    LDS-DMA load still in flight will write, and lands every load at once: both ends of the race)."""
    seg_len, n_blocks = 8, 200
    got, want, n_out, peak, w, text = _run_wave(oracle, 9000 - 100 * q_first, True, seg_len, n_blocks, 70 + q_first + d, shift, D=d, q_first=q_first, extra={"X5_DEEP": "1"})
    assert np.array_equal(got[:n_blocks].view(U), want[:n_blocks].view(U)), np.nonzero(got[:n_blocks].view(U) != want[:n_blocks].view(U))[0][:8]
    assert (got[n_blocks:] == F(3.0e8)).all()
    assert "s_waitcnt vmcnt(8)" in text and "s_waitcnt vmcnt(4)" not in text


@pytest.mark.parametrize("d,q_first,shift", [(16, 32, -100), (16, 0, 0), (8, 32, -140), (8, 20, -4), (4, 32, -204), (4, 8, -12)])
def test_exact5_load_tiles_requested_in_pairs(oracle, d, q_first, shift):
    """X5_BURST=1: two load tiles per request slot (256 contiguous bytes of a stream at 192 kHz)."""
    seg_len, n_blocks = 8, 200
    got, want, n_out, peak, w, text = _run_wave(oracle, 9000 - 100 * q_first, True, seg_len, n_blocks, 75 + q_first + d, shift, D=d, q_first=q_first, extra={"X5_BURST": "1"})
    assert np.array_equal(got[:n_blocks].view(U), want[:n_blocks].view(U)), np.nonzero(got[:n_blocks].view(U) != want[:n_blocks].view(U))[0][:8]
    assert (got[n_blocks:] == F(3.0e8)).all()


def test_exact5_inc_file_is_the_generators_output():
    assert _gen() == open(os.path.join(ROOT, "cwsl_digi_amd", "csrc", "exact5_asm.inc")).read(), "exact5_asm.inc is not the generator's output"
