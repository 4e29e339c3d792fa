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
        if n >= 1:
            # the last accumulation of the step before (its products of sample 15), kept back so that it does not sit two instructions
            # behind the accumulation of sample 14 on the same register: here the eleven load instructions above separate them
            e("v_pk_add_f32 %s, %s, %s" % (v2(sxp), v2(sxp), v2(XA)))
            e("v_pk_add_f32 %s, %s, %s" % (v2(syp), v2(syp), v2(YA)))
        # one software pipeline over the sixteen samples: products alternate between two register sets, every accumulation four
        # instructions behind its product; the previous step's four-operation tail is dropped into the first gaps
        MX(sx, 0); MY(sy, 0)
        MX(XA, 1); MY(YA, 1)
        if tail:
            e("v_pk_mul_f32 %s, %s, %s" % (v2(TA), v2(sxp), v2(php)))
        sets = [(XB, YB), (XA, YA)]                       # sample m (>= 2) uses sets[m % 2]; sample 1 used (XA, YA)
        for m in range(2, 16):
            px, py = sets[m % 2]
            MX(px, m); MY(py, m)
            ox, oy = sets[(m - 1) % 2]
            AX(ox); AY(oy)
            if tail and m == 2:
                e("v_pk_mul_f32 %s, %s, %s op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" % (v2(TB), v2(syp), v2(php)))
            if tail and m == 4:
                e("v_pk_add_f32 %s, %s, %s" % (v2(TA), v2(TA), v2(TB)))
                if n == 1:
                    e("v_mov_b32 v%d, 0" % (TA + 1))      # step 0's tail: tap block -1 does not exist (output o0 + 1 gets +0)
            if tail and m == 6:
                e("v_pk_add_f32 %[w], %[w], " + v2(TA))
        assert sets[15 % 2] == (XA, YA)                   # sample 15's products: accumulated at the top of the next step
    # step 32's last accumulation, then its tail: tap block 32 does not exist (output o0 gets +0)
    sx, sy, ph = SX[0], SY[0], PH[32 % 3]
    e("s_nop 1")
    e("v_pk_add_f32 %s, %s, %s" % (v2(sx), v2(sx), v2(XA)))
    e("v_pk_add_f32 %s, %s, %s" % (v2(sy), v2(sy), v2(YA)))
    e("s_nop 3")
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
