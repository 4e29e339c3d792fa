#!/usr/bin/env python3
"""Generates cwsl_digi_amd/csrc/lab/exact3_asm.inc: the 33 FIR steps of demod_exact3_kernel<16, ...> as ONE assembly statement.

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

    python scripts/gen_exact3_asm.py > cwsl_digi_amd/csrc/lab/exact3_asm.inc
"""
VBASE = {16: 160, 8: 100, 4: 76}     # first fixed VGPR: high enough to leave the compiler its own registers (next tile's prefetch: 68 / 36 / 20),
                                       # low enough for three (96 kHz) and four (48 kHz) waves per SIMD where the LDS image allows them
SBASE = 36                            # first fixed SGPR


def regmap(D):
    b = VBASE[D]
    m = dict(QA=b, QB=b + 2 * D)      # sample registers of the two buffers (2 D each)
    t = b + 4 * D
    m["PH"] = [t, t + 2, t + 4]       # block phase, three in rotation (current, previous for the tail, next in flight)
    m["SX"] = [t + 6, t + 10]         # running sums by step parity
    m["SY"] = [t + 8, t + 12]
    m["XA"], m["YA"], m["XB"], m["YB"], m["TA"], m["TB"] = t + 14, t + 16, t + 18, t + 20, t + 22, t + 24
    m["VTOP"] = t + 26
    m["HA"], m["HB"] = SBASE, SBASE + 2 * D      # tap rows (2 D SGPRs each)
    m["STOP"] = SBASE + 4 * D
    return m


def row_bytes(D):
    return (D + 2) * 8       # LDS row pitch: D samples + phase + pad (16-byte aligned rows)


def v2(r):
    return "v[%d:%d]" % (r, r + 1)


def s2(r):
    return "s[%d:%d]" % (r, r + 1)


def gen(D):
    ROW, TROW = row_bytes(D), 2 * D * 4
    R = regmap(D)
    QA, QB, PH, SX, SY, HA, HB = R["QA"], R["QB"], R["PH"], R["SX"], R["SY"], R["HA"], R["HB"]
    XA, YA, XB, YB, TA, TB = R["XA"], R["YA"], R["XB"], R["YB"], R["TA"], R["TB"]
    out = []
    e = out.append

    def issue(n):
        h = HA if n % 2 == 0 else HB
        q = QA if n % 2 == 0 else QB
        row = "%[r0]" if n % 2 == 0 else "%[r1]"
        off = (n >> 1) * ROW
        if D == 16:
            e("s_load_dwordx16 s[%d:%d], %%[tp], 0x%x" % (h, h + 15, n * TROW))
            e("s_load_dwordx16 s[%d:%d], %%[tp], 0x%x" % (h + 16, h + 31, n * TROW + 64))
        elif D == 8:
            e("s_load_dwordx16 s[%d:%d], %%[tp], 0x%x" % (h, h + 15, n * TROW))
        else:
            e("s_load_dwordx8 s[%d:%d], %%[tp], 0x%x" % (h, h + 7, n * TROW))
        for k in range(D // 2):
            e("ds_read_b128 v[%d:%d], %s offset:%d" % (q + 4 * k, q + 4 * k + 3, row, off + 16 * k))
        e("ds_read_b64 %s, %s offset:%d" % (v2(PH[n % 3]), row, off + 8 * D))

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
        for m in range(2, D):
            px, py = sets[m % 2]
            MX(px, m); MY(py, m)
            ox, oy = sets[(m - 1) % 2]
            AX(ox); AY(oy)
            if tail and m == 2:
                e("v_pk_mul_f32 %s, %s, %s op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" % (v2(TB), v2(syp), v2(php)))
            if tail and m == (4 if D >= 8 else 3):
                if D < 8:
                    e("s_nop 1")
                e("v_pk_add_f32 %s, %s, %s" % (v2(TA), v2(TA), v2(TB)))
                if n == 1:
                    e("v_mov_b32 v%d, 0" % (TA + 1))      # step 0's tail: tap block -1 does not exist (output o0 + 1 gets +0)
            if tail and D >= 8 and m == 6:
                e("v_pk_add_f32 %[w], %[w], " + v2(TA))
        if tail and D < 8:
            e("s_nop 1")
            e("v_pk_add_f32 %[w], %[w], " + v2(TA))
        assert sets[(D - 1) % 2] == (XA, YA)              # the last sample's products: accumulated at the top of the next step
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


print("// GENERATED by scripts/gen_exact3_asm.py -- do not edit.  See that script and demod_exact3_kernel (demod_kernels.hpp).")
print("#define EXACT3_ASM_ROW_BYTES(D) (((D) + 2) * 8)")
for D in (16, 8, 4):
    R = regmap(D)
    print("#define EXACT3_ASM_CLOBBERS_%d " % D + ", ".join('"v%d"' % r for r in range(VBASE[D], R["VTOP"])) + ", "
          + ", ".join('"s%d"' % r for r in range(SBASE, R["STOP"])) + ', "memory"')
    print("#define EXACT3_FIR%d_ASM \\" % D)
    print(" \\\n".join('    "%s\\n\\t"' % l for l in gen(D)))
    print()
