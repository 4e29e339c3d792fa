#!/usr/bin/env python3
"""Generates cwsl_digi_amd/csrc/exact5_asm.inc: the whole life of one wave of demod_exact5_kernel (192 kHz, round 5) as ONE assembly statement.

Why a new shape.  demod_exact4_kernel pays for the reference's arithmetic at 1.76 GHz: lane = output, so every block row of the tile image is
re-read from LDS by 17 lanes (1.15 MB per 512 outputs) and the chip sits at its package power limit.  Round 5's micro-benchmark
(scripts/micro/mfma_k1.hip, profiles/r5_mfma_k1.txt) measured the same arithmetic with every operand in registers at 980-1110 ns per 1024 sums
and 2.38 GHz, against exact4's 1570 ns.  Here therefore LANE = STREAM and time runs along the lane:

  * a wave serves 32 consecutive segments ("streams") of ONE channel's pending blocks: lane j and lane 32 + j both belong to stream j; a lane of
    the lower half carries Re, of the upper half Im, of the mixed sample y = in * tone (SSBD.hpp:167) and the tap blocks n = 0..15 / 16..31;
  * per block ("tile": 16 samples of each stream): y[m] = in.re * c1[m] + in.im * c2[m] (c = (tone.re, -tone.im) in the lower half, (tone.im,
    tone.re) in the upper: the un-fused complex product's Re and Im), then for m = 0..15 ONE v_mfma_f32_32x32x1_2b_f32 with A = lane i -> h[m +
    16 n(i)], B = y[m], C = 0: the 2048 products fl(y * h) of all 32 tap blocks, Re and Im (K = 1: one rounding per product, bit-identical to
    v_mul_f32 -- measured on 2^32 pairs), landing in the lane of their stream; S = P_0, S = fl(S + P_m) in the reference's order (:168);
  * T_r = S.re_r * k1 + S.im_r * k2 -- the component of sum * phase (:170) that Iterate() will read of the output this term belongs to (Re for
    even outputs, Im for odd ones: :131-134), the block's phase kept and advanced in the lane (phase *= phase_inc, :174);
  * the 32-term accumulation of an output (:170, oldest block first) runs down a 17-register shift chain: W[r + 1] = W[r] + T_r (tap block r of the
    lower half, 16 + r of the upper); once per tile the lower half's finished partial (n = 0..15) crosses to the upper half (v_permlane32_swap)
    where it collects n = 16..31, and the upper half's W[16] is the finished output;
  * the wave streams its 32 x 128 bytes per tile through LDS only to transpose them (coalesced 16-byte pieces, eight lanes per line -> one row per
    stream, read back lane = stream); nothing is re-read: 12 KB of LDS traffic per 1024 sums where exact4 moves 72 KB.  Product form (X5_DMA=1): the
    pieces go from the ring straight into LDS (global_load_lds_dwordx4), four buffers per wave; X5_DMA=0 is the round's first form (staging registers
    and ds_write_b128, two buffers).  The ordered sums are issued as register pairs (X5_PK=1: v_pk_add_f32, each half rounded on its own).
A stream starts 32 blocks before its first output (the workspace warm-up: an output needs its 32 blocks); where that reaches before the
demodulator's origin the rows are zeros and the phase is held at (1, 0) (lds_write, phase_hold).

Zeros.  The MFMA computes fmaf(a, b, +0): an exact-zero product arrives as +0 where v_mul_f32 gives -0.  That cannot change an output: the
reference's running sum starts as (+0 + p_0) and is therefore never -0, so adding either zero leaves it unchanged; where a whole sum is zero its
term T is a zero of either sign, and the workspace slot it is added to -- which itself starts as +0 + T_0 -- is never -0 either.

    python scripts/gen_exact5_asm.py > cwsl_digi_amd/csrc/exact5_asm.inc

tests/test_exact5_stream.py runs this text on a 64-lane emulator (tests/wave_emulator.py) against the oracle's demodulator and compares bits; it also
re-derives every wait the text needs (MFMA results, LDS and memory loads) and fails on a read that is not covered."""
import re
import sys

# ---- register map (all fixed; the statement's operands are extra registers chosen by the compiler) ----
S, DA, DB, H, Y, IN, C1, C2, W = 0, 32, 64, 96, 112, 128, 144, 160, 176
NPIM, P, STG, OB, PEAK, T = 193, 194, 196, 212, 216, 217           # P = (re, im) at 194/195; temporaries T..T+7
VTOP = 225
import os
LOADER_IN_GAP = os.environ.get("X5_LOADER_IN_GAP", "1") == "1"
GROUPED = os.environ.get("X5_GROUPED", "1") == "1"            # mix and sum * phase as groups of independent instructions (0: the first form's short dependent chains)
# the ring is read once: non-temporal (same-box A/B: 2.5-2.8 % of the launch).  X5_RING_NT: "0" none, "1" nt, or the modifiers themselves ("sc1 nt", ...)
_nt = os.environ.get("X5_RING_NT", "1")
RING_NT = "" if _nt == "0" else (" nt" if _nt == "1" else " " + _nt.replace("_", " ").strip())
# X5_DMA=1: the rows go from HBM straight into LDS (global_load_lds_dwordx4: no staging registers, no ds_write); the LDS image is then lane-linear
# (32 unpadded rows of 128 bytes per load tile, four buffers), conflict-free through a swizzle on the SOURCE side: lane l of a load fetches piece
# (l & 7) ^ f(row) of its row, f(row) = (row >> 1) & 7, and lane j reads piece k of its row from slot k ^ f(j).
DMA = os.environ.get("X5_DMA", "1") == "1"
# X5_DEEP=1 (LDS-DMA form only): THREE load tiles in flight instead of two, in the same four buffers -- nothing has to be done when rows arrive, so the wait for
# load tile t + 1 moves from the start of tile t to just before its first read (near the end of tile t), and tile t + 3 is requested before it, not after
DEEP = DMA and os.environ.get("X5_DEEP", "0") == "1"
# X5_BURST=1 (LDS-DMA form only): load tiles are requested in PAIRS (every second load tile: 256 contiguous bytes of every stream at 192 kHz instead of 128 --
# half as many DRAM row activations), waits as in the DEEP form
BURST = DMA and os.environ.get("X5_BURST", "0") == "1"
PK = os.environ.get("X5_PK", "1") == "1"                      # the block sums as v_pk_add_f32 pairs: same issue time, less power (same-box: clock 2.00 -> 2.05-2.08 GHz, -1.2 %)
AK, ZQ = 196, 204                                                  # DMA form: v196-v203 the eight read addresses of the lane's row, v204-v207 a quad of zeros
DBUF = 4096                                                        # DMA form: bytes of one load tile in LDS (32 rows x 128)
D = 16                                                             # samples per block = Fs / 12 kHz: 16 (192 kHz), 8 (96 kHz), 4 (48 kHz); set by program(d)
STG2 = 240                                                         # the second staging set, at the top of the file (the operands sit between VTOP and it)
ROW = 144                                                          # LDS row pitch: 16 samples + 16 bytes (conflict-free 16-byte reads, lane = row)
BUF = 32 * ROW                                                     # one tile of one wave: 4608 bytes; two buffers
MFMA_WAIT = 18                                                     # issue slots between a 16-pass f32 MFMA and a VALU access to its result
WARM_ITERS = 8                                                     # 32 warm-up tiles


def geometry():
    """R compute tiles share one 128-byte row ("load tile": 16 samples of every stream, whatever D); an iteration is four load tiles = NT compute
    tiles; a tile's samples arrive in NSUB reads of SUB samples each."""
    R = 16 // D
    return R, 4 * R, min(D, 8), D // min(D, 8)


def mfma(dst, m):
    return f"v_mfma_f32_32x32x1_2b_f32 v[{dst}:{dst + 31}], v{H + m}, v{Y + m}, 0"


def adds(src):
    if PK:                                                          # X5_PK=1: the sums as register pairs (two independently rounded additions per instruction: the same bits)
        return [f"v_pk_add_f32 v[{S + k}:{S + k + 1}], v[{S + k}:{S + k + 1}], v[{src + k}:{src + k + 1}]" for k in range(0, 32, 2)]
    return [f"v_add_f32 v{S + k}, v{S + k}, v{src + k}" for k in range(32)]


def mix(m):
    mm, t = m % min(D, 8), T + 2 * (m & 3)
    return [f"v_mul_f32 v{t}, v{IN + 2 * mm}, v{C1 + m}", f"v_mul_f32 v{t + 1}, v{IN + 2 * mm + 1}, v{C2 + m}", f"v_add_f32 v{Y + m}, v{t}, v{t + 1}"]


def mix_group(ms):
    """Four samples at a time, the eight products first: no instruction waits for the one just before it (a wave issues a dependent VALU
    instruction later than an independent one; its partner wave fills only some of those slots)."""
    out = []
    if not GROUPED:
        return [x for m in ms for x in mix(m)]
    for g in range(0, len(ms), 4):
        grp = [mix(m) for m in ms[g:g + 4]]
        out += [x for tri in grp for x in tri[:2]] + [tri[2] for tri in grp]
    return out


def sub_read(c, sub):
    """Sub-read `sub` of compute tile c of the iteration (c may be NT: the next iteration's tile 0): SUB samples from the tile's part of its row."""
    R, NT, SUB, NSUB = geometry()
    lt = c // R                                                    # load tile; its LDS buffer is lt & 1 (four load tiles per iteration: the parity carries over)
    if DMA:                                                        # four buffers, one per load tile of the iteration; piece k of the row through address register k
        k0 = ((c % R) * 8 * D + sub * 8 * SUB) // 16
        return [f"ds_read_b128 v[{IN + 4 * k}:{IN + 4 * k + 3}], v{AK + k0 + k} offset:{(lt & 3) * DBUF}" for k in range(SUB // 2)]
    off = (lt & 1) * BUF + (c % R) * 8 * D + sub * 8 * SUB
    return [f"ds_read_b128 v[{IN + 4 * k}:{IN + 4 * k + 3}], %[ldsr] offset:{off + 16 * k}" for k in range(SUB // 2)]


def stg(tile):
    return STG2 if tile & 1 else STG                                # tile t's rows travel through staging set t & 1 into LDS buffer t & 1


_uid = [0]


def lds_write(tile):
    """The staged rows go to LDS -- after the rows that lie BEFORE the demodulator's origin have been replaced by zeros (x[i < 0] = 0, SSBD.hpp:117-121: a
    fresh SSBD has a zero workspace).  Only the first wave of a channel whose first pending output is one of the demodulator's first 32 has such rows
    (%[holdlt] load tiles of them at most: a wave-uniform count; every other wave pays the compare and the branch).  A loader lane keeps the count of its
    FIRST stream's pre-origin load tiles (%[tapoff], recycled after the prologue); its i-th stream starts 8 i streams = %[st<i>] load tiles later."""
    _uid[0] += 1
    k = _uid[0]
    if DMA:                                                        # the rows are in LDS already: overwrite the pre-origin ones there (loader lane l owns bytes 16 l .. of each load)
        z = ["s_cmp_eq_u32 %[holdlt], 0", f"s_cbranch_scc1 L5_Z{k}_%=", "s_sub_u32 %[holdlt], %[holdlt], 1"]
        for i in range(4):
            z += [f"v_cmp_lt_i32 vcc, {'0' if i == 0 else '%%[st%d]' % i}, %[tapoff]", "s_and_saveexec_b64 %[esave], vcc",
                  f"ds_write_b128 %[ldsw], v[{ZQ}:{ZQ + 3}] offset:{(tile & 3) * DBUF + 1024 * i}", "s_mov_b64 exec, %[esave]"]
        return z + ["v_add_u32 %[tapoff], -1, %[tapoff]", f"L5_Z{k}_%=:"]
    z = ["s_cmp_eq_u32 %[holdlt], 0", f"s_cbranch_scc1 L5_Z{k}_%=", "s_sub_u32 %[holdlt], %[holdlt], 1"]
    for i in range(4):
        z += [f"v_cmp_lt_i32 vcc, {'0' if i == 0 else '%%[st%d]' % i}, %[tapoff]"]
        z += [f"v_cndmask_b32_e64 v{stg(tile) + 4 * i + j}, v{stg(tile) + 4 * i + j}, 0, vcc" for j in range(4)]
    z += ["v_add_u32 %[tapoff], -1, %[tapoff]", f"L5_Z{k}_%=:"]
    return z + [f"ds_write_b128 %[ldsw], v[{stg(tile) + 4 * i}:{stg(tile) + 4 * i + 3}] offset:{(tile & 1) * BUF + 8 * ROW * i}" for i in range(4)]


def phase_hold():
    """phase = (1, 0) at the origin (SSBD.hpp:121): while a stream's blocks precede it (%[ckoff] tiles left, per lane; %[hold] = the wave's largest) its
    phase is put back to (1, 0) after every step -- what it multiplied zeros with in between does not matter."""
    _uid[0] += 1
    k = _uid[0]
    return ["s_cmp_eq_u32 %[hold], 0", f"s_cbranch_scc1 L5_H{k}_%=", "s_sub_u32 %[hold], %[hold], 1", "v_cmp_lt_i32 vcc, 0, %[ckoff]",
            f"v_cndmask_b32_e64 v{P}, v{P}, 1.0, vcc", f"v_cndmask_b32_e64 v{P + 1}, v{P + 1}, 0, vcc", f"v_mul_f32 v{NPIM}, -1.0, v{P + 1}",
            "v_add_u32 %[ckoff], -1, %[ckoff]", f"L5_H{k}_%=:"]


def ring_loads(tile):
    """The four 16-byte loads of a wave for one load tile (32 rows of 128 bytes).  192 kHz: a stream's start and the ring's length are multiples of 512
    bytes (64 samples = 4 blocks: the push granularity), so four load tiles share one offset (immediates 0 / 128 / 256 / 384) and the offsets advance --
    and wrap -- once per iteration.  96 / 48 kHz: the guaranteed alignment is 4 blocks = 256 / 128 bytes, so the offsets advance after every load tile."""
    imm = 128 * (tile & 3) if D == 16 else 0
    if DMA:                                                        # M0 = where the load's 1024 bytes go; the immediate would move BOTH addresses, so 192 kHz uses four ring bases
        L = []
        for i in range(4):
            L += [f"s_add_u32 m0, %[ldsb], {(tile & 3) * DBUF + 1024 * i}", "s_nop 0",
                  f"global_load_lds_dwordx4 %[off{i}], %[ring{(tile & 3) if D == 16 else 0}]{RING_NT}"]
        return L if D == 16 else L + ["s_nop 0"] + advance_offsets()
    L = [f"global_load_dwordx4 v[{stg(tile) + 4 * i}:{stg(tile) + 4 * i + 3}], %[off{i}], %[ring] offset:{imm}{RING_NT}" for i in range(4)]
    return L if D == 16 else L + ["s_nop 0"] + advance_offsets()


def advance_offsets():
    step = "0x200" if D == 16 else "0x80"
    out = []
    for i in range(4):                                             # the next 512 / 128 bytes of each stream; the ring's end is a multiple of that away from a stream's start
        par = str(i & 1) if DMA else ""                                # (DMA form: the swizzled piece of a lane differs between even and odd loads)
        out += [f"v_add_u32 %[off{i}], {step}, %[off{i}]", f"v_cmp_eq_u32 vcc, %[off{i}], %[capl{par}]", f"v_cndmask_b32 %[off{i}], %[off{i}], %[pc16{par}], vcc"]
    return out


def t_and_w(u):
    """sum * phase, the component the output will read, and the workspace chain (descending r: every add reads the slot's previous occupant).
    The 32 products go to the product buffer DB, the 16 terms to DA -- both are free between the last sums of a tile and the next tile's MFMA 1 --
    so that every group is 16 or 32 independent instructions."""
    out = []
    if not GROUPED:
        for r in range(15, -1, -1):
            k1, k2 = (P, NPIM) if (u + r) & 1 else (P + 1, P)
            t = T + 3 * (r & 1)
            out += [f"v_mul_f32 v{t}, v{S + r}, v{k1}", f"v_mul_f32 v{t + 1}, v{S + 16 + r}, v{k2}", f"v_add_f32 v{t + 2}, v{t}, v{t + 1}",
                    f"v_add_f32 v{W + r + 1}, v{W + r}, v{t + 2}"]
        return out + [f"v_mov_b32 v{W}, 0"]
    for r in range(16):
        re_type = (u + r) & 1                                       # output b = q + 31 - n is even: Re(sum * phase) = S.re p.re - S.im p.im
        k1, k2 = (P, NPIM) if re_type else (P + 1, P)               # odd: Im = S.re p.im + S.im p.re
        out += [f"v_mul_f32 v{DB + r}, v{S + r}, v{k1}", f"v_mul_f32 v{DB + 16 + r}, v{S + 16 + r}, v{k2}"]
    out += [f"v_add_f32 v{DA + r}, v{DB + r}, v{DB + 16 + r}" for r in range(16)]
    out += [f"v_add_f32 v{W + r + 1}, v{W + r}, v{DA + r}" for r in range(15, -1, -1)]
    # the lower half's partial (tap blocks 0..15 done) becomes the upper half's W[0]; the lower half starts a fresh slot (+0)
    out += [f"v_mov_b32 v{W}, 0"]
    return out


def swap_and_out(u):
    g = ["v_mov_b32 v{d}, v{s}", "v_mul_f32 v{d}, %[nsign], v{s}", "v_mul_f32 v{d}, -1.0, v{s}", "v_mul_f32 v{d}, %[sign], v{s}"][u]   # Iterate(): +Re, -Im sign, -Re, +Im sign
    return [f"v_permlane32_swap_b32 v{W}, v{W + 16}", g.format(d=OB + u, s=W + 16)]


def phase_step():
    t = T
    return [f"v_mul_f32 v{t}, %[incre], v{P}", f"v_mul_f32 v{t + 1}, %[incim], v{P + 1}", f"v_mul_f32 v{t + 2}, %[incim], v{P}", f"v_mul_f32 v{t + 3}, %[incre], v{P + 1}",
            f"v_sub_f32 v{P}, v{t}, v{t + 1}", f"v_add_f32 v{P + 1}, v{t + 2}, v{t + 3}", f"v_mul_f32 v{NPIM}, -1.0, v{P + 1}"]


def tile(c):
    """Compute tile c of an iteration (c mod 4 = the block's position mod 4).  Memory pipeline, per LOAD tile (16 samples of every stream): TWO are in
    flight (one per wave was 8 MB in flight chip-wide: at ~2.8 us of loaded HBM latency that is the 2.9 TB/s the first form of this kernel ran at,
    whatever its arithmetic did).  In the first compute tile of load tile t: wait for load tile t + 1's rows (the older of the two outstanding
    sets: vmcnt(4)), zero what lies before the origin (register form: move them to LDS buffer (t + 1) & 1), and request load tile t + 3 into the
    buffer (staging set) that load tile t - 1 has left.  X5_DEEP / X5_BURST: measured alternatives of this schedule (profiles/r5_experiments.txt)."""
    R, NT, SUB, NSUB = geometry()
    u = c & 3
    # DEEP: the next compute tile's first samples come from the NEXT load tile -- its rows must have landed (three load tiles outstanding: vmcnt(8); a store
    # among them only makes the wait stricter), and its pre-origin rows are zeroed, just before that read
    arrive = []
    if (DEEP or BURST) and (c + 1) % R == 0:
        n = (c + 1) // R                                            # the load tile about to be read; BURST: an even one was requested together with n + 1, an odd one
        arrive = [f"s_waitcnt vmcnt({8 if DEEP or n & 1 else 4})"] + lds_write(n)   # with n - 1, and the pair n + 1, n + 2 has been requested since
    L = ["s_waitcnt lgkmcnt(0)"]                                    # this tile's first SUB samples (read during the previous tile)
    L += mix_group(list(range(SUB)))
    L += sub_read(c, 1) if NSUB == 2 else arrive + sub_read(c + 1, 0)   # the rest of this tile's samples / the next tile's, into the same registers
    loader = []
    if c % R == 0:
        lt = c // R
        if not (DEEP or BURST):
            loader = ["s_waitcnt vmcnt(4)"]                         # the NEXT load tile's rows have arrived from the ring ...
            loader += lds_write(lt + 1)                             # ... transposed through LDS ...
        if c % 4 == 0:
            loader += store_block(c // 4)                           # the previous four outputs (behind the older loads, ahead of the new ones: see vmcnt)
        if BURST:
            if lt % 2 == 0:
                if lt == 2 and D == 16:
                    loader += advance_offsets()
                loader += ring_loads(lt + 2) + ring_loads(lt + 3)
        else:
            if lt == 1 and D == 16:
                loader += advance_offsets()
            loader += ring_loads(lt + 3)                            # ... and the third load tile from here is requested
    elif c % 4 == 0:
        raise AssertionError("a store slot that is not a loader slot")
    if not LOADER_IN_GAP:
        L += loader
    where = {}
    for m in range(D):
        dst = S if m == 0 else (DA if m & 1 else DB)
        L.append(mfma(dst, m))
        where[m] = dst
        if m == 1 and LOADER_IN_GAP:                                # MFMA 1's products are 18 issue slots away: the memory pipeline's instructions wait here
            L += loader
        if m >= 2:
            L += adds(where[m - 1])
        if m == 5 and NSUB == 2:                                    # samples 8..15 have long landed: mix them before MFMA 8 needs y[8]
            L += ["s_waitcnt lgkmcnt(0)"]
            L += mix_group(list(range(8, 16)))
            L += arrive + sub_read(c + 1, 0)                        # the next tile's samples 0..7
    L += adds(where[D - 1])
    L += t_and_w(u)
    L += phase_step()                                               # (two instructions at least between W[0]'s write and the swap that reads it)
    L += phase_hold()
    L += swap_and_out(u)
    return L


def prologue():
    L = ["s_nop 4"]                                                 # operands fresh from v_readfirstlane are read as addresses below (5 wait states)
    L += [f"global_load_dwordx4 v[{H + 4 * k}:{H + 4 * k + 3}], %[tapoff], %[taps] offset:{16 * k}" for k in range(D // 4)]
    L += [f"global_load_dwordx2 v[{P}:{P + 1}], %[ckoff], %[ckpt]"]
    L += {16: ["s_load_dwordx16 s[64:79], %[tone], 0x0", "s_load_dwordx16 s[80:95], %[tone], 0x40"], 8: ["s_load_dwordx16 s[64:79], %[tone], 0x0"],
          4: ["s_load_dwordx8 s[64:71], %[tone], 0x0"]}[D]
    L += ring_loads(0) + ring_loads(1)
    L += [f"v_mov_b32 v{W + k}, 0" for k in range(17)] + [f"v_mov_b32 v{OB + k}, 0" for k in range(4)] + [f"v_mov_b32 v{PEAK}, 0"]
    L += ["s_waitcnt vmcnt(0) lgkmcnt(0)"]
    # the two address operands have done their work: from here on they count pre-origin tiles (this lane's stream) / load tiles (this loader lane's first stream)
    L += ["v_and_b32 %[ckoff], 0xffff, %[pk]", "v_lshrrev_b32 %[tapoff], 16, %[pk]"]
    L += ["s_mov_b64 vcc, %[hmask]"]
    for m in range(D):                                              # c1 = (tone.re | tone.im), c2 = (-tone.im | tone.re) by half
        L += [f"v_mov_b32 v{T}, s{64 + 2 * m}", f"v_mov_b32 v{T + 1}, s{65 + 2 * m}", f"v_cndmask_b32 v{C1 + m}, v{T}, v{T + 1}, vcc",
              f"v_xor_b32 v{T + 2}, 0x80000000, v{T + 1}", f"v_cndmask_b32 v{C2 + m}, v{T + 2}, v{T}, vcc"]
    L += [f"v_mul_f32 v{NPIM}, -1.0, v{P + 1}"]
    if DMA:                                                         # slot of piece k in this lane's row: k ^ f(row); %[fj16] = f(row) << 4
        for k in range(8):
            L += [f"v_xor_b32 v{AK + k}, {16 * k}, %[fj16]", f"v_add_u32 v{AK + k}, %[ldsr], v{AK + k}"]
        L += [f"v_mov_b32 v{ZQ + j}, 0" for j in range(4)]
    L += lds_write(0)
    if not BURST:
        L += ring_loads(2)
    L += sub_read(0, 0)
    return L


def store_block(k):
    """Four outputs per lane, 16 bytes: the streams of the upper half whose range is not exhausted; skipped for the first nine slots (the 32 warm-up
    tiles' outputs and the slot before the first tile)."""
    tag = "E" if k == "E" else str(k)
    return ["s_cmp_gt_u32 %[warm], 0", f"s_cbranch_scc1 L5_WARM{tag}_%=",
            "v_cmp_lt_i32 vcc, 0, %[rem]", "s_and_b64 vcc, vcc, %[hmask]", "s_and_saveexec_b64 %[esave], vcc",
            f"global_store_dwordx4 %[outoff], v[{OB}:{OB + 3}], %[out]"] + \
           [f"v_max_f32 v{PEAK}, v{PEAK}, |v{OB + j}|" for j in range(4)] + \
           ["s_mov_b64 exec, %[esave]", "v_add_u32 %[outoff], 16, %[outoff]", "v_add_u32 %[rem], -4, %[rem]", f"s_branch L5_NEXT{tag}_%=",
            f"L5_WARM{tag}_%=:", "s_sub_u32 %[warm], %[warm], 1", f"L5_NEXT{tag}_%=:"]


def regs_of(tok):
    tok = tok.strip().strip("|")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def split_ops(l):
    op, _, rest = l.partition(" ")
    rest = re.sub(r"\s+offset:\d+", "", rest)
    return op, ([t.strip() for t in re.split(r",\s*(?![^\[]*\])", rest)] if rest else [])


def slots(L):
    """Issue slots of the SHORTEST path through L: what lies between a forward branch and its target may be skipped and counts nothing."""
    n, skip_to = 0, None
    for l in L:
        if skip_to is not None:
            if l == skip_to + ":":
                skip_to = None
            continue
        if l.endswith(":"):
            continue
        n += int(l.split()[1]) + 1 if l.startswith("s_nop") else 1
        if l.startswith(("s_cbranch", "s_branch")):
            skip_to = l.split()[1]
    return n


def fix_hazards(L):
    """hipcc pads nothing inside an asm string.  Rules applied (the f32 MFMAs are 16-pass, not XDL): an MFMA's result may be touched by the VALU
    MFMA_WAIT issue slots after the MFMA at the earliest -- or once two later MFMAs have issued (they complete in order); a VALU result is an MFMA
    operand two slots later at the earliest; v_permlane32_swap reads a VALU result two slots later at the earliest."""
    out, pend, wrote, n_mfma = [], [], {}, 0
    for l in L:
        op, toks = split_ops(l)
        used = set().union(*[regs_of(t) for t in toks]) if toks else set()
        need = 0
        if op.startswith(("v_", "ds_write", "global_store")):
            for (regs, pos, idx) in pend:
                if regs & used and n_mfma - idx < (1 if op.startswith("v_mfma") else 2):
                    need = max(need, MFMA_WAIT - (slots(out) - pos))
        if op.startswith("v_mfma") or op.startswith("v_permlane"):
            for t in (toks[1:3] if op.startswith("v_mfma") else toks[0:2]):
                for r in regs_of(t):
                    if r in wrote:
                        need = max(need, 2 - (slots(out) - wrote[r]))
        if need > 0:
            out.append(f"s_nop {need - 1}")
        if op.startswith("v_mfma"):
            n_mfma += 1
            pend = [(r, p, i) for (r, p, i) in pend if n_mfma - i < 3]
            pend.append((regs_of(toks[0]), slots(out) + 1, n_mfma))
        elif op.startswith("v_") and toks and not op.startswith("v_cmp"):
            for r in regs_of(toks[0]):
                wrote[r] = slots(out) + 1
        out.append(l)
    return out


def program(d):
    global D
    D = d
    _uid[0] = 0
    R, NT, SUB, NSUB = geometry()
    pro = fix_hazards(prologue())
    if DMA:
        pro = ["s_mov_b32 %[m0keep], m0"] + pro
    body = []
    for c in range(NT):
        body += tile(c)
    body = fix_hazards(body + [mfma(S, 0)])[:-1]                    # (the loop wraps: the hazards of the first MFMA against the body's end hold as well)
    loop = ["L5_LOOP_%=:"] + body + ["s_sub_u32 %[iters], %[iters], 1", "s_cmp_lg_u32 %[iters], 0", "s_cbranch_scc1 L5_LOOP_%="]
    epi = store_block("E") + ["s_waitcnt vmcnt(0) lgkmcnt(0)", f"v_mov_b32 %[peak], v{PEAK}"] + (["s_mov_b32 m0, %[m0keep]"] if DMA else [])
    return pro, loop, epi


def main():
    w = sys.stdout.write
    w("// GENERATED by scripts/gen_exact5_asm.py -- do not edit.  See that script and demod_exact5_kernel (demod_kernels.hpp).\n")
    w(f"#define EXACT5_ASM_DMA {int(DMA)}\n")
    w(f"#define EXACT5_ASM_ROW_BYTES {128 if DMA else ROW}\n#define EXACT5_ASM_BUF_BYTES {DBUF if DMA else BUF}\n#define EXACT5_ASM_NBUF {4 if DMA else 2}\n"
      f"#define EXACT5_ASM_WARM_STORES 9\n#define EXACT5_ASM_VTOP {VTOP}\n")
    w("#define EXACT5_ASM_CLOBBERS " + ", ".join(f'"v{i}"' for i in list(range(VTOP)) + ([] if DMA else list(range(STG2, STG2 + 16)))) + ", " +
      ", ".join(f'"s{i}"' for i in range(64, 96)) + ', "vcc", "scc", "memory"\n')
    for d in (16, 8, 4):
        pro, loop, epi = program(d)
        w(f"#define EXACT5_D{d}_TILES_PER_ITER {geometry()[1]}\n")
        for name, lines in ((f"EXACT5_D{d}_PROLOGUE_ASM", pro), (f"EXACT5_D{d}_LOOP_ASM", loop), (f"EXACT5_D{d}_EPILOGUE_ASM", epi)):
            w(f"#define {name} \\\n")
            for l in lines:
                w(f'    "{l}\\n\\t" \\\n')
            w('    ""\n\n')


if __name__ == "__main__":
    main()
