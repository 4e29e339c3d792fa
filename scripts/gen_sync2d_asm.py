#!/usr/bin/env python3
"""Generates cwsl_digi_amd/csrc/sync2d_asm.inc: the hand-scheduled LDS read / add pipeline of ft8_sync2d_v3_kernel.

Per bin a wave needs 42 ds_read_b64 for the Costas sums (21 from the band image, 21 from its 7-tone sums c0) and 21 more to
build the NEXT bin's c0.  hipcc issues them in batches closed by lgkmcnt(0) (eight LDS round trips per bin); here they are
one in-order stream with nine reads in flight, every add waiting only for its own operand (lgkmcnt(N) counted from the
issue order), temporaries in fixed registers v94..v127 (declared as clobbers).  The sums are the restatement's, in its
order: acc = x0; acc = acc + x1; ... (0 + x0 = x0 exactly: the powers are never -0).

    python scripts/gen_sync2d_asm.py > cwsl_digi_amd/csrc/sync2d_asm.inc
"""
import os
HOIST = os.environ.get("X2D_HOIST", "1") == "1"   # 0: the third-column address of the c0 build recomputed for each of its seven rows (round 3's form; A/B: scripts/gpu_r5_x2d.sh)
RB = 378 * 4                      # bytes per row of the band image (S2_PITCH floats)
ICOS = [3, 1, 4, 0, 6, 5, 2]
RING0 = 104                       # v104..v127: 12 ring temporaries (v2f each)
KACC0 = 98                        # v98..v103: accumulators of the three c0 columns
ADDR0 = 94                        # v94, v95: persistent unclamped bases; v96, v97: rotating clamped-address temporaries
DEPTH = 3                         # units (3 reads each) in flight behind the one being consumed: twelve ring temporaries, twelve reads outstanding


def reg2(base):
    return "v[%d:%d]" % (base, base + 1)


class Gen:
    def __init__(self):
        self.lines = []
        self.issued = 0            # reads issued so far
        self.done = 0              # reads known to have landed
        self.rot = 0

    def emit(self, s):
        self.lines.append(s)

    def read(self, dst, addr, off):
        assert self.issued - self.done < 15, "the LDS counter of a wave holds 15"
        self.emit("ds_read_b64 %s, %s%s" % (dst, addr, (" offset:%d" % off) if off else ""))
        self.issued += 1
        return self.issued         # 1-based index of this read in the in-order stream

    def wait_for(self, idx):
        if idx <= self.done:
            return                 # an earlier wait already covered it (LDS returns in order)
        n = self.issued - idx      # reads issued after it may still be outstanding
        assert 0 <= n <= 15
        self.emit("s_waitcnt lgkmcnt(%d)" % n)
        self.done = idx

    def clamped(self, sbase, voff):
        t = "v%d" % (ADDR0 + 2 + self.rot)
        self.rot ^= 1
        self.emit("v_add_u32 %s, %s, %s" % (t, sbase, voff))
        return t


def search_units():
    """14 units of three reads: (n, 't') = va, vb, vc of Costas symbol n from the band image; (n, 'u') = wa, wb, wc from c0."""
    units = []
    for n in range(7):
        for ty in "tu":
            units.append(("S", n, ty))
    return units


def build(with_search, with_c0):
    g = Gen()
    if with_search or with_c0:
        pass
    XS, XC, XN = "v%d" % ADDR0, "v%d" % (ADDR0 + 1), None
    if with_search:
        g.emit("v_add_u32 %s, %%[sS], %%[vU]" % XS)
    g.emit("v_add_u32 %s, %%[sC], %%[vU]" % XC)
    units = (search_units() if with_search else []) + ([("K", k, "") for k in range(7)] if with_c0 else [])
    ring = 0
    XL = [None]
    pending = []                  # (unit, [(read index, temp reg)]) issued, adds not yet emitted
    acc = {"t": ["%[ta]", "%[tb]", "%[tc]"], "u": ["%[ua]", "%[ub]", "%[uc]"]}

    def issue(u):
        nonlocal ring, XN
        kind, n, ty = u
        regs = []
        if kind == "S":
            direct = n == 0
            row = 2 * ICOS[n] * RB
            for which in range(3):                                    # a, b, c
                if direct:
                    dst = acc[ty][which]
                else:
                    dst = reg2(RING0 + 2 * ring); ring = (ring + 1) % 12
                    assert not any(dst == r for _, rs in pending for _, r in rs), "ring temporary still holds an unconsumed operand"
                base_s, base_x = ("%[sS]", XS) if ty == "t" else ("%[sC]", XC)
                rowoff = row if ty == "t" else 0
                if which == 0:
                    a = g.clamped(base_s, "%%[vA%d]" % n); off = rowoff
                elif which == 1:
                    a = base_x; off = rowoff + (48 + 2 * n) * 8
                elif n < 4:
                    a = base_x; off = rowoff + (120 + 2 * n) * 8
                else:
                    a = g.clamped(base_s, "%%[vC%d]" % n); off = rowoff
                regs.append((g.read(dst, a, off), dst))
        else:
            k = n
            if k == 0:
                XN = "v%d" % ADDR0                                      # the search no longer needs its band base: reuse the register
                g.emit("v_add_u32 %s, %%[sN], %%[vU]" % XN)
            for col in range(3):
                if k == 0:
                    dst = reg2(KACC0 + 2 * col)
                else:
                    dst = reg2(RING0 + 2 * ring); ring = (ring + 1) % 12
                    assert not any(dst == r for _, rs in pending for _, r in rs), "ring temporary still holds an unconsumed operand"
                if col < 2:
                    a = XN; off = 2 * k * RB + 512 * col
                else:
                    if k == 0 or not HOIST:                             # one address for the third column of all seven rows (no search unit is issued after this:
                        XL[0] = g.clamped("%[sN]", "%[vL2]")            # the rotating temporary stays untouched until the stream's last write)
                    a = XL[0]; off = 2 * k * RB
                regs.append((g.read(dst, a, off), dst))
        pending.append((u, regs))

    def consume():
        u, regs = pending.pop(0)
        kind, n, ty = u
        g.wait_for(regs[-1][0])                                       # LDS returns in order: the unit's last read covers all three
        if n == 0:
            return                                                    # read straight into its accumulator: nothing to add
        for which, (idx, reg) in enumerate(regs):
            if kind == "S":
                a = acc[ty][which]
            else:
                a = reg2(KACC0 + 2 * which)
            g.emit("v_pk_add_f32 %s, %s, %s" % (a, a, reg))

    for u in units:
        issue(u)
        while len(pending) > DEPTH:
            consume()
    while pending:
        consume()
    if with_c0:
        # the accumulators of unit K0 were read directly: make sure they have landed (they were issued before everything consumed above)
        g.emit("ds_write_b64 %s, %s" % (XC, reg2(KACC0)))
        g.emit("ds_write_b64 %s, %s offset:512" % (XC, reg2(KACC0 + 2)))
        t = g.clamped("%[sC]", "%[vL2]")
        g.emit("ds_write_b64 %s, %s" % (t, reg2(KACC0 + 4)))
    g.emit("s_waitcnt lgkmcnt(0)")
    return g.lines


def cstring(lines):
    return "\n".join('    "%s\\n\\t"' % l for l in lines)


print("// GENERATED by scripts/gen_sync2d_asm.py -- do not edit.  See that script and ft8_sync2d_v3_kernel (sync_kernels.hpp).")
print("#define SYNC2D_ASM_CLOBBERS " + ", ".join('"v%d"' % r for r in range(ADDR0, 128)) + ', "memory"')
for name, s, c in (("SYNC2D_ASM_SEARCH_NEXT", True, True), ("SYNC2D_ASM_SEARCH_LAST", True, False), ("SYNC2D_ASM_C0_ONLY", False, True)):
    print("#define %s \\" % name)
    body = build(s, c)
    print(" \\\n".join('    "%s\\n\\t"' % l for l in body))
    print()
