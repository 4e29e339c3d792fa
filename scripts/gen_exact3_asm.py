#!/usr/bin/env python3
"""Generates cwsl_digi_amd/csrc/exact3_asm.inc: the 33 FIR steps of demod_exact3_kernel<16, ...> as ONE assembly statement.

The C++ form of this loop (rounds 2-3) had hipcc in the way at every turn: loads sunk to their first use, waits hoisted, a pad
instruction at every statement boundary, and -- what this file is for -- no way to name the upper pair of a 128-bit register
quad from inline assembly, so the mixed samples were fetched with seventeen 8-byte LDS reads per step.  With every register fixed
(v160..v249, s36..s99, declared as clobbers) a step reads its sixteen samples with eight ds_read_b128 and the packed operations
name the pairs directly: 79 instead of 87 instructions per step at the ~5 cycles per instruction one wave issues.

Arithmetic and order are those of the C++ form: per step and sample m
    sX = (t.x, t.x) * (h[m + 16 n], h[m + 16 (n - 1)])  [m = 0 starts the sum]   sX += ... [m > 0]      (SSBD.hpp:167-168, un-fused)
    sY likewise with t.y
and one step later the tail   W += sX * (ph.x, ph.y) + sY * (-ph.y, ph.x)   (:170), whose first (step 0) and last (step 32)
instances drop the component of the tap block that does not exist.

    python scripts/gen_exact3_asm.py > cwsl_digi_amd/csrc/exact3_asm.inc
"""
D = 16
ROW = (D + 2) * 8            # LDS row pitch in bytes: 16 samples + phase + pad (16-byte aligned rows)
TROW = 2 * D * 4             # bytes per tap row
QA, QB = 160, 192            # sample registers of the two buffers (32 each)
PH = [224, 226, 228]         # block phase, three in rotation (current, previous for the tail, next in flight)
SX = [230, 234]              # running sums by step parity
SY = [232, 236]
XA, YA, XB, YB, TA, TB = 238, 240, 242, 244, 246, 248
HA, HB = 36, 68              # tap rows (32 SGPRs each)


def v2(r):
    return "v[%d:%d]" % (r, r + 1)


def s2(r):
    return "s[%d:%d]" % (r, r + 1)


def gen():
    out = []
    e = out.append

    def issue(n):
        h = HA if n % 2 == 0 else HB
        q = QA if n % 2 == 0 else QB
        row = "%[r0]" if n % 2 == 0 else "%[r1]"
        off = (n >> 1) * ROW
        e("s_load_dwordx16 s[%d:%d], %%[tp], 0x%x" % (h, h + 15, n * TROW))
        e("s_load_dwordx16 s[%d:%d], %%[tp], 0x%x" % (h + 16, h + 31, n * TROW + 64))
        for k in range(8):
            e("ds_read_b128 v[%d:%d], %s offset:%d" % (q + 4 * k, q + 4 * k + 3, row, off + 16 * k))
        e("ds_read_b64 %s, %s offset:%d" % (v2(PH[n % 3]), row, off + 128))

    issue(0)
    for n in range(33):
        h = HA if n % 2 == 0 else HB
        q = QA if n % 2 == 0 else QB
        sx, sy = SX[n % 2], SY[n % 2]
        sxp, syp, php = SX[(n - 1) % 2], SY[(n - 1) % 2], PH[(n - 1) % 3]
        e("s_waitcnt lgkmcnt(0)")
        if n < 32:
            issue(n + 1)

        def MX(p, m, first=False):
            e("v_pk_mul_f32 %s, %s, %s op_sel_hi:[1,0]" % (v2(p), s2(h + 2 * m), v2(q + 2 * m)))

        def MY(p, m):
            e("v_pk_mul_f32 %s, %s, %s op_sel:[0,1]" % (v2(p), s2(h + 2 * m), v2(q + 2 * m)))

        def AX(p):
            e("v_pk_add_f32 %s, %s, %s" % (v2(sx), v2(sx), v2(p)))

        def AY(p):
            e("v_pk_add_f32 %s, %s, %s" % (v2(sy), v2(sy), v2(p)))

        tail = n >= 1
        MX(sx, 0); MY(sy, 0)
        if tail:
            e("v_pk_mul_f32 %s, %s, %s" % (v2(TA), v2(sxp), v2(php)))
        MX(XB, 1); MY(YB, 1)
        if tail:
            e("v_pk_mul_f32 %s, %s, %s op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" % (v2(TB), v2(syp), v2(php)))
        MX(XA, 2); MY(YA, 2); AX(XB); AY(YB)
        if tail:
            e("v_pk_add_f32 %s, %s, %s" % (v2(TA), v2(TA), v2(TB)))
            if n == 1:
                e("v_mov_b32 v%d, 0" % (TA + 1))          # step 0's tail: tap block -1 does not exist (output o0 + 1 gets +0)
        MX(XB, 3); MY(YB, 3); AX(XA); AY(YA)
        if tail:
            e("v_pk_add_f32 %[w], %[w], " + v2(TA))
        MX(XA, 4); MY(YA, 4); AX(XB); AY(YB)
        MX(XB, 5); MY(YB, 5); AX(XA); AY(YA)
        MX(XA, 6); MY(YA, 6); AX(XB); AY(YB)
        MX(XB, 7); MY(YB, 7); AX(XA); AY(YA)
        AX(XB); AY(YB)
        MX(XA, 8); MY(YA, 8); MX(XB, 9); MY(YB, 9); AX(XA); AY(YA)
        MX(XA, 10); MY(YA, 10); AX(XB); AY(YB)
        MX(XB, 11); MY(YB, 11); AX(XA); AY(YA)
        MX(XA, 12); MY(YA, 12); AX(XB); AY(YB)
        MX(XB, 13); MY(YB, 13); AX(XA); AY(YA)
        MX(XA, 14); MY(YA, 14); AX(XB); AY(YB)
        MX(XB, 15); MY(YB, 15); AX(XA); AY(YA)
        AX(XB); AY(YB)
    # the tail of step 32: tap block 32 does not exist (output o0 gets +0)
    sx, sy, ph = SX[0], SY[0], PH[32 % 3]
    e("v_pk_mul_f32 %s, %s, %s" % (v2(TA), v2(sx), v2(ph)))
    e("v_pk_mul_f32 %s, %s, %s op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" % (v2(TB), v2(sy), v2(ph)))
    e("s_nop 3")
    e("v_pk_add_f32 %s, %s, %s" % (v2(TA), v2(TA), v2(TB)))
    e("s_nop 3")
    e("v_mov_b32 v%d, 0" % TA)
    e("s_nop 1")
    e("v_pk_add_f32 %[w], %[w], " + v2(TA))
    return out


lines = gen()
print("// GENERATED by scripts/gen_exact3_asm.py -- do not edit.  See that script and demod_exact3_kernel (demod_kernels.hpp).")
print("#define EXACT3_ASM_ROW_BYTES %d" % ROW)
print("#define EXACT3_ASM_CLOBBERS " + ", ".join('"v%d"' % r for r in range(160, 250)) + ", " + ", ".join('"s%d"' % r for r in range(36, 100)) + ', "memory"')
print("#define EXACT3_FIR16_ASM \\")
print(" \\\n".join('    "%s\\n\\t"' % l for l in lines))
