#!/usr/bin/env python3
"""Generates scripts/micro/mfma_k1_loop.inc: straight-line tile bodies for mfma_k1.hip (the go / no-go micro-benchmark of round 5).

One "tile" is what demod_exact5 would do for one 16-sample block of 32 streams (lane = stream, lanes 0-31 Re / 32-63 Im of the mixed
signal): 16 x v_mfma_f32_32x32x1_2b_f32 (the 2048 un-fused products fl(y*h) of SSBD.hpp:167-168 each), 15 x 16 v_pk_add_f32 (the ordered
sums), and the block's remaining scalar work (mix 48, sum*phase 48, workspace 16, phasor 6).  Variants:
   mfma_only   16 MFMAs, nothing else
   valu_only   the VALU work, no MFMA
   valu_all    the VALU work with the products as 16 x 16 v_pk_mul_f32 (what demod_exact4_kernel issues per 1024 sums, without its LDS traffic)
   both_pk     MFMAs with the packed adds of the previous product set in each gap
   both_sc     the same with 32 v_add_f32 instead of 16 v_pk_add_f32 per gap
   gapN        16 MFMAs with N independent v_pk_add_f32 per gap (N = 0, 4, 8, 12, 16, 20): the price of a filler beside this MFMA
Registers (fixed): v0-31 S | v32-63 DA | v64-95 DB | v96-111 h | v112-127 y | v128-159 in | v160-191 c1,c2 | v192-208 W | v209-223 misc."""
import re
import sys

S, DA, DB, H, Y, IN, C1, C2, W = 0, 32, 64, 96, 112, 128, 160, 176, 192
P, PI, PC, T0 = 209, 211, 213, 217          # phase, phase_inc, 4 phase coefficients, temporaries 217..223
MFMA_WAIT = 18                              # wait states between a 16-pass SGEMM MFMA and a VALU access to its D


def mfma(dst, m):
    return f"v_mfma_f32_32x32x1_2b_f32 v[{dst}:{dst + 31}], v{H + m}, v{Y + m}, 0"


def pk_adds(dst, src, n=16, first=0):
    return [f"v_pk_add_f32 v[{dst + 2 * k}:{dst + 2 * k + 1}], v[{dst + 2 * k}:{dst + 2 * k + 1}], v[{src + 2 * k}:{src + 2 * k + 1}]" for k in range(first, first + n)]


def sc_adds(dst, src):
    return [f"v_add_f32 v{dst + k}, v{dst + k}, v{src + k}" for k in range(32)]


def mix():
    out = []
    for m in range(16):
        out += [f"v_mul_f32 v{T0}, v{IN + 2 * m}, v{C1 + m}", f"v_mul_f32 v{T0 + 1}, v{IN + 2 * m + 1}, v{C2 + m}", f"v_add_f32 v{Y + m}, v{T0}, v{T0 + 1}"]
    return out


def phase_and_ws():
    out = []
    for r in range(16):                                     # T_r = S.re_r * p1 + S.im_r * p2 (coefficient pair by the parity of r)
        c = PC + 2 * (r & 1)
        out += [f"v_mul_f32 v{T0 + 2}, v{S + r}, v{c}", f"v_mul_f32 v{T0 + 3}, v{S + 16 + r}, v{c + 1}", f"v_add_f32 v{T0 + 4 + (r & 1)}, v{T0 + 2}, v{T0 + 3}"]
        out += [f"v_add_f32 v{W + 16 - r}, v{W + 15 - r}, v{T0 + 4 + (r & 1)}"]
    out += [f"v_mov_b32 v{W}, 0"]
    # phase *= phase_inc, un-fused
    out += [f"v_mul_f32 v{T0}, v{P}, v{PI}", f"v_mul_f32 v{T0 + 1}, v{P + 1}, v{PI + 1}", f"v_mul_f32 v{T0 + 2}, v{P}, v{PI + 1}", f"v_mul_f32 v{T0 + 3}, v{P + 1}, v{PI}",
            f"v_sub_f32 v{P}, v{T0}, v{T0 + 1}", f"v_add_f32 v{P + 1}, v{T0 + 2}, v{T0 + 3}"]
    return out


def lds_reads():
    return [f"ds_read_b128 v[{IN + 4 * k}:{IN + 4 * k + 3}], %[la] offset:{16 * k}" for k in range(8)]


def tile(variant):
    L = []
    if variant == "mfma_only":
        for m in range(16):
            L.append(mfma([S, DA, DB][m % 3] if m else S, m))
        return L
    if variant.startswith("gap"):
        n = int(variant[3:])
        for m in range(16):
            L.append(mfma(DA if m & 1 else DB, m))
            L += [f"v_pk_add_f32 v[{S + 2 * (k % 16)}:{S + 2 * (k % 16) + 1}], v[{S + 2 * (k % 16)}:{S + 2 * (k % 16) + 1}], v[{IN + 2 * (k % 16)}:{IN + 2 * (k % 16) + 1}]" for k in range(n)]
        return L
    if variant.startswith("sgap"):
        n = int(variant[4:])
        for m in range(16):
            L.append(mfma(DA if m & 1 else DB, m))
            L += [f"v_add_f32 v{S + (k % 32)}, v{S + (k % 32)}, v{IN + (k % 32)}" for k in range(n)]
        return L
    if variant == "both_sc1":
        # ONE product buffer: MFMA m -> DA, its 18 wait slots filled with the block's scalar work (a sixteenth of it per gap), the 32 adds, MFMA m + 1.
        # MFMA 0 writes S itself.  (The f32 MFMA occupies the SIMD's FP32 lanes -- measured: MFMA and VALU time add up -- so a second buffer hides nothing.)
        L.append("s_waitcnt lgkmcnt(0)")
        other = mix() + phase_and_ws()
        per = (len(other) + 15) // 16
        L += lds_reads()
        for m in range(16):
            L.append(mfma(S if m == 0 else DA, m))
            L += other[m * per:(m + 1) * per]
            if m:
                L += sc_adds(S, DA)
        return L
    with_mfma = variant not in ("valu_only", "valu_all")
    adds = sc_adds if variant == "both_sc" else pk_adds
    # tile start: the samples have landed; mix, refill, then the MFMA chain with the previous set's sums in each gap
    L.append("s_waitcnt lgkmcnt(0)")
    L += mix()
    L += lds_reads()
    L.append("s_nop 1")
    bufs = [S, DA, DB]
    where = {}
    for m in range(16):
        dst = S if m == 0 else (DA if m & 1 else DB)
        if with_mfma:
            L.append(mfma(dst, m))
        elif variant == "valu_all":                          # the products on the VALU: 16 packed multiplies stand for one MFMA's 2048 products
            L += [f"v_pk_mul_f32 v[{dst + 2 * k}:{dst + 2 * k + 1}], v[{IN + 2 * k}:{IN + 2 * k + 1}], v[{C1 + 2 * k}:{C1 + 2 * k + 1}]" for k in range(16)]
        where[m] = dst
        if m >= 2:
            L += adds(S, where[m - 1])
    L += adds(S, where[15])
    L += phase_and_ws()
    return L


def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def fix_hazards(L):
    """s_nop padding so that every access to an MFMA's D registers comes >= MFMA_WAIT issue slots after that MFMA (hipcc pads nothing
    inside an asm string); a VALU result read by an MFMA as A/B needs 2."""
    out, pend, age_w, n_mfma = [], [], {}, 0
    for l in L:
        op, _, rest = l.partition(" ")
        toks = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", rest)] if rest else []
        used = set().union(*[regs_of(t) for t in toks]) if toks else set()
        need = 0
        for (regs, pos, idx) in pend:                       # MFMAs complete in order: once two later ones have issued, this one's D is written
            if regs & used and n_mfma - idx < (1 if op.startswith("v_mfma") else 2):
                need = max(need, MFMA_WAIT - (len_slots(out) - pos))
        if op.startswith("v_mfma"):
            for t in toks[1:3]:
                for r in regs_of(t):
                    if r in age_w:
                        need = max(need, 2 - (len_slots(out) - age_w[r]))
        if need > 0:
            out.append(f"s_nop {need - 1}")
        if op.startswith("v_mfma"):
            pend = [(r, p, i) for (r, p, i) in pend if len_slots(out) - p < MFMA_WAIT]
            n_mfma += 1
            pend.append((regs_of(toks[0]), len_slots(out) + 1, n_mfma))
        elif op.startswith("v_") and toks:
            for r in regs_of(toks[0]):
                age_w[r] = len_slots(out) + 1
        out.append(l)
    return out


def len_slots(L):
    n = 0
    for l in L:
        n += int(l.split()[1]) + 1 if l.startswith("s_nop") else 1
    return n


def main():
    variants = ["mfma_only", "valu_only", "valu_all", "both_pk", "both_sc", "both_sc1", "gap0", "gap4", "gap8", "gap12", "gap16", "gap20", "sgap16", "sgap32"]
    with open(sys.argv[1], "w") as f:
        f.write("// GENERATED by scripts/micro/gen_mfma_k1_loop.py -- do not edit.\n")
        for v in variants:
            body = fix_hazards(tile(v))
            f.write(f"#define K1_TILE_{v.upper()} \\\n")
            for l in body:
                f.write(f'    "{l}\\n\\t" \\\n')
            f.write('    ""\n')
            f.write(f"#define K1_NINST_{v.upper()} {len(body)}\n")
        f.write("#define K1_VARIANTS(X) " + " ".join(f"X({v.upper()})" for v in variants) + "\n")
        f.write('#define K1_CLOBBERS ' + ", ".join(f'"v{i}"' for i in range(224)) + "\n")


if __name__ == "__main__":
    main()
