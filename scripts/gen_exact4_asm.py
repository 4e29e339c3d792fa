#!/usr/bin/env python3
"""Generates cwsl_digi_amd/csrc/lab/exact4_asm.inc: the FIR of demod_exact4_kernel (192 kHz) as TWO assembly streams, one per half of the 33 steps.

Round 4.  demod_exact3_kernel keeps a pair of adjacent outputs on one lane for all 33 steps: 250 registers per lane and a 78 KB LDS image per
four waves, i.e. two waves per SIMD -- and scripts/micro/pk_issue.hip shows what that costs: a SIMD retires a packed FP32 operation every 5.2
cycles from one wave, 4.46 from two, 4.25 from four (exact3's average: 4.95).  The only dependence between the steps of an output is the
final accumulation  W = ((0 + R_0) + R_1) + ... + R_32  of the per-block terms R_n = sum_n * phase_n (SSBD.hpp:170); the R_n themselves are
independent.  So the steps are split between TWO waves that share the tile's image:

    stream A (waves 0-3 of the workgroup, lane = output pair): steps 0..16, accumulating W_A = R_0 + ... + R_16 as exact3 does;
    stream B (waves 4-7, same pairs):                          steps 17..32, KEEPING R_17..R_32 in registers; then, behind the workgroup
                                                               barrier at which A has published W_A in LDS, W = (W_A + R_17) + ... + R_32
                                                               -- the reference's additions in the reference's order.

Eight waves per 78 KB image = four waves per SIMD, so every stream must fit 128 registers: the samples of a step live in ONE buffer of 32
registers that is refilled half a step ahead (samples 0-7 are re-read for step n + 1 while samples 8-15 of step n are being used, and vice
versa: two `s_waitcnt lgkmcnt(0)` per step, each draining reads issued half a step earlier), the block phase rotates through two register
pairs, and B's sixteen R_n take 32 more.  Arithmetic and its order are exact3's (gen_exact3_asm.py), operation for operation.

    python scripts/gen_exact4_asm.py > cwsl_digi_amd/csrc/lab/exact4_asm.inc
"""
D = 16
SPLIT = 17                 # stream A: steps 0..16, stream B: steps 17..32
VCOMMON = 72               # v72..v127: the registers both streams use
VR = 40                    # v40..v71: stream B's R_17..R_32
SBASE = 36                 # s36..s99: two tap rows
ROW, TROW = (D + 2) * 8, 2 * D * 4


def regmap():
    b = VCOMMON
    m = dict(Q=b)                       # 32 sample registers: samples 0-7 (first half of a step), 8-15 (second half)
    t = b + 2 * D
    m["PH"] = [t, t + 2]               # block phase of the step being computed / of the step before (for its tail)
    m["SX"] = [t + 4, t + 8]
    m["SY"] = [t + 6, t + 10]
    m["XA"], m["YA"], m["XB"], m["YB"], m["TA"], m["TB"] = t + 12, t + 14, t + 16, t + 18, t + 20, t + 22
    m["VTOP"] = t + 24
    assert m["VTOP"] == 128
    m["R"] = [VR + 2 * k for k in range(33 - SPLIT)]
    assert m["R"][-1] + 2 == VCOMMON
    m["H"] = [SBASE, SBASE + 2 * D]
    m["STOP"] = SBASE + 4 * D
    return m


def v2(r):
    return "v[%d:%d]" % (r, r + 1)


def s2(r):
    return "s[%d:%d]" % (r, r + 1)


def gen(stream):
    R = regmap()
    Q, PH, SX, SY, H = R["Q"], R["PH"], R["SX"], R["SY"], R["H"]
    XA, YA, XB, YB, TA, TB = R["XA"], R["YA"], R["XB"], R["YB"], R["TA"], R["TB"]
    n0, n1 = (0, SPLIT - 1) if stream == "A" else (SPLIT, 32)
    out = []
    e = out.append

    def row(n):
        return ("%[r0]" if n % 2 == 0 else "%[r1]"), (n >> 1) * ROW

    def taps(n):
        h = H[n % 2]
        e("s_load_dwordx16 s[%d:%d], %%[tp], 0x%x" % (h, h + 15, n * TROW))
        e("s_load_dwordx16 s[%d:%d], %%[tp], 0x%x" % (h + 16, h + 31, n * TROW + 64))

    def load_lo(n):                    # samples 0-7 of step n
        r, off = row(n)
        for k in range(D // 4):
            e("ds_read_b128 v[%d:%d], %s offset:%d" % (Q + 4 * k, Q + 4 * k + 3, r, off + 16 * k))

    def load_hi(n):                    # samples 8-15 of step n and its block phase
        r, off = row(n)
        for k in range(D // 4, D // 2):
            e("ds_read_b128 v[%d:%d], %s offset:%d" % (Q + 4 * k, Q + 4 * k + 3, r, off + 16 * k))
        e("ds_read_b64 %s, %s offset:%d" % (v2(PH[n % 2]), r, off + 8 * D))

    def tail_result(n):                # where the tail of step n (= R_n) goes: A accumulates it into W, B keeps it
        return None if stream == "A" else R["R"][n - SPLIT]

    taps(n0)
    load_lo(n0)
    for n in range(n0, n1 + 1):
        h = H[n % 2]
        sx, sy = SX[n % 2], SY[n % 2]
        sxp, syp, php = SX[(n - 1) % 2], SY[(n - 1) % 2], PH[(n - 1) % 2]
        tail = n > n0                  # the tail of step n - 1 rides in this step's first gaps

        def MX(p, m):
            e("v_pk_mul_f32 %s, %s, %s op_sel_hi:[1,0]" % (v2(p), s2(h + 2 * m), v2(Q + 2 * m)))

        def MY(p, m):
            e("v_pk_mul_f32 %s, %s, %s op_sel:[0,1]" % (v2(p), s2(h + 2 * m), v2(Q + 2 * m)))

        def AX(p):
            e("v_pk_add_f32 %s, %s, %s" % (v2(sx), v2(sx), v2(p)))

        def AY(p):
            e("v_pk_add_f32 %s, %s, %s" % (v2(sy), v2(sy), v2(p)))

        e("s_waitcnt lgkmcnt(0)")      # samples 0-7 and the taps of this step (issued half a step / a step ago); nothing newer is in flight
        load_hi(n)                     # samples 8-15 were last read by the step before: refill them now, needed half a step from here
        if n < n1:
            taps(n + 1)
        if tail:
            e("v_pk_add_f32 %s, %s, %s" % (v2(sxp), v2(sxp), v2(XA)))     # the step before: its last accumulation (sample 15)
            e("v_pk_add_f32 %s, %s, %s" % (v2(syp), v2(syp), v2(YA)))
        MX(sx, 0); MY(sy, 0)
        MX(XA, 1); MY(YA, 1)
        if tail:
            e("v_pk_mul_f32 %s, %s, %s" % (v2(TA), v2(sxp), v2(php)))
        sets = [(XB, YB), (XA, YA)]
        for m in range(2, D):
            if m == D // 2:
                e("s_waitcnt lgkmcnt(0)")          # samples 8-15 (issued at the top of the step) and the next step's taps
                if n < n1:
                    load_lo(n + 1)                 # samples 0-7 have been consumed: refill them for the next step
            px, py = sets[m % 2]
            MX(px, m); MY(py, m)
            ox, oy = sets[(m - 1) % 2]
            AX(ox); AY(oy)
            if tail and m == 2:
                e("v_pk_mul_f32 %s, %s, %s op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" % (v2(TB), v2(syp), v2(php)))
            if tail and m == 4:
                dst = tail_result(n - 1)
                e("v_pk_add_f32 %s, %s, %s" % (v2(TA if dst is None else dst), v2(TA), v2(TB)))
                if n == 1:
                    e("v_mov_b32 v%d, 0" % (TA + 1))      # step 0's tail: tap block -1 does not exist (output o0 + 1 gets +0)
            if tail and m == 6 and stream == "A":
                e("v_pk_add_f32 %[w], %[w], " + v2(TA))
        assert sets[(D - 1) % 2] == (XA, YA)
    # the last step of the stream: its last accumulation, then its tail
    sx, sy, ph = SX[n1 % 2], SY[n1 % 2], PH[n1 % 2]
    e("s_nop 1")
    e("v_pk_add_f32 %s, %s, %s" % (v2(sx), v2(sx), v2(XA)))
    e("v_pk_add_f32 %s, %s, %s" % (v2(sy), v2(sy), v2(YA)))
    e("s_nop 3")
    e("v_pk_mul_f32 %s, %s, %s" % (v2(TA), v2(sx), v2(ph)))
    e("v_pk_mul_f32 %s, %s, %s op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" % (v2(TB), v2(sy), v2(ph)))
    e("s_nop 3")
    if stream == "A":
        e("v_pk_add_f32 %s, %s, %s" % (v2(TA), v2(TA), v2(TB)))
        e("s_nop 3")
        e("v_pk_add_f32 %[w], %[w], " + v2(TA))
    else:
        last = R["R"][32 - SPLIT]
        e("v_pk_add_f32 %s, %s, %s" % (v2(last), v2(TA), v2(TB)))
        e("s_nop 3")
        e("v_mov_b32 v%d, 0" % last)                       # step 32's tail: tap block 32 does not exist (output o0 gets +0)
        # every wave of the workgroup meets here: stream A has published W_A (one ds_write_b64 per lane, drained before its barrier)
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        e("ds_read_b64 %[w], %[xa]")
        e("s_waitcnt lgkmcnt(0)")
        for k in range(33 - SPLIT):                        # W = (W_A + R_17) + R_18 + ... + R_32, in the reference's order
            e("v_pk_add_f32 %[w], %[w], " + v2(R["R"][k]))
    return out


if __name__ == "__main__":
    R = regmap()
    print("// GENERATED by scripts/gen_exact4_asm.py -- do not edit.  See that script and demod_exact4_kernel (demod_kernels.hpp).")
    print("#define EXACT4_ASM_ROW_BYTES %d" % ROW)
    print("#define EXACT4_ASM_SPLIT %d" % SPLIT)
    sregs = ", ".join('"s%d"' % r for r in range(SBASE, R["STOP"]))
    print("#define EXACT4_ASM_CLOBBERS_A " + ", ".join('"v%d"' % r for r in range(VCOMMON, R["VTOP"])) + ", " + sregs + ', "memory"')
    print("#define EXACT4_ASM_CLOBBERS_B " + ", ".join('"v%d"' % r for r in range(VR, R["VTOP"])) + ", " + sregs + ', "memory"')
    for stream in "AB":
        print("#define EXACT4_FIR%s_ASM \\" % stream)
        print(" \\\n".join('    "%s\\n\\t"' % l for l in gen(stream)))
        print()
