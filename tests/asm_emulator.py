"""A one-lane emulator of the instructions the generated FIR streams use (scripts/gen_exact3_asm.py, gen_exact4_asm.py), in IEEE float32:
enough to run a stream on the CPU against a numpy restatement of ProcessBlock's arithmetic (SSBD.hpp:160-183) and compare bits.

VOP3P packed-f32 semantics (gfx950): dst.lo = src0[op_sel[0]] (*|+) src1[op_sel[1]], dst.hi = src0[op_sel_hi[0]] (*|+) src1[op_sel_hi[1]];
defaults op_sel = [0, 0], op_sel_hi = [1, 1]; neg_lo / neg_hi negate the selected source of the low / high result."""
import re

import numpy as np

F = np.float32


class Lane:
    def __init__(self, lds_bytes, tap_bytes, r0, r1, xa=0):
        self.v = {}                                  # VGPR -> float32 bit pattern (np.uint32)
        self.s = {}
        self.lds = np.frombuffer(bytes(lds_bytes), dtype=np.uint32).copy()
        self.taps = np.frombuffer(bytes(tap_bytes), dtype=np.uint32)
        self.ph = {"r0": r0, "r1": r1, "xa": xa}
        self.w = [np.uint32(0), np.uint32(0)]
        self.barriers = 0
        self.on_barrier = None                       # callback run at s_barrier (the other stream's publication)

    def _get(self, op):
        op = op.strip()
        if op == "%[w]":
            return list(self.w)
        m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", op)
        bank = self.v if m.group(1) == "v" else self.s
        return [bank[int(m.group(2))], bank[int(m.group(3))]]

    def _put(self, op, lo, hi):
        op = op.strip()
        if op == "%[w]":
            self.w = [lo, hi]
            return
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
        self.v[int(m.group(1))], self.v[int(m.group(2))] = lo, hi

    @staticmethod
    def _mods(text):
        mods = {"op_sel": [0, 0], "op_sel_hi": [1, 1], "neg_lo": [0, 0], "neg_hi": [0, 0]}
        for k, a, b in re.findall(r"(op_sel_hi|op_sel|neg_lo|neg_hi):\[(\d),(\d)\]", text):
            mods[k] = [int(a), int(b)]
        return mods

    def run(self, lines):
        for l in lines:
            op, _, rest = l.partition(" ")
            if op in ("s_waitcnt", "s_nop"):
                continue
            if op == "s_barrier":
                self.barriers += 1
                if self.on_barrier:
                    self.on_barrier(self)
                continue
            if op.startswith("s_load_dwordx"):
                n = int(op[len("s_load_dwordx"):])
                m = re.fullmatch(r"s\[(\d+):(\d+)\], %\[tp\], 0x([0-9a-f]+)", rest.strip())
                base, off = int(m.group(1)), int(m.group(3), 16)
                assert int(m.group(2)) - base + 1 == n and off % 4 == 0
                for k in range(n):
                    self.s[base + k] = self.taps[off // 4 + k]
                continue
            if op in ("ds_read_b128", "ds_read_b64"):
                n = 4 if op.endswith("128") else 2
                m = re.fullmatch(r"(v\[(\d+):(\d+)\]|%\[w\]), %\[(r0|r1|xa)\](?: offset:(\d+))?", rest.strip())
                assert m, l
                addr = self.ph[m.group(4)] + int(m.group(5) or 0)
                assert addr % (4 * n) == 0 or n == 2 and addr % 8 == 0, l
                vals = [self.lds[addr // 4 + k] for k in range(n)]
                if m.group(1) == "%[w]":
                    self.w = vals
                else:
                    base = int(m.group(2))
                    assert int(m.group(3)) - base + 1 == n
                    for k in range(n):
                        self.v[base + k] = vals[k]
                continue
            if op == "v_mov_b32":
                m = re.fullmatch(r"v(\d+), 0", rest.strip())
                self.v[int(m.group(1))] = np.uint32(0)
                continue
            if op in ("v_pk_mul_f32", "v_pk_add_f32"):
                ops_text = re.split(r"\s+(?=op_sel|neg_)", rest.strip(), maxsplit=1)
                dst, a, b = [x.strip() for x in ops_text[0].split(",")]
                md = self._mods(ops_text[1] if len(ops_text) > 1 else "")
                A = np.array(self._get(a), np.uint32).view(F)
                B = np.array(self._get(b), np.uint32).view(F)
                res = []
                for sel, neg in ((md["op_sel"], md["neg_lo"]), (md["op_sel_hi"], md["neg_hi"])):
                    x, y = F(A[sel[0]]), F(B[sel[1]])
                    if neg[0]: x = F(-x)
                    if neg[1]: y = F(-y)
                    res.append(F(x * y) if op == "v_pk_mul_f32" else F(x + y))
                r = np.array(res, F).view(np.uint32)
                self._put(dst, r[0], r[1])
                continue
            raise AssertionError("emulator: unknown instruction: " + l)
        return np.array(self.w, np.uint32).view(F)
