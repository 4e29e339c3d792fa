"""A 64-lane emulator of the instructions scripts/gen_exact5_asm.py emits (one wave of demod_exact5_kernel), in IEEE float32 on numpy: enough to run
the generated text on the CPU -- prologue, loop with its branches, epilogue -- against the oracle's demodulator and compare bits, and to re-derive the
waits the text needs: a register that a memory / LDS load has not yet delivered (no covering s_waitcnt) or that an MFMA has not yet written (fewer
than 18 issue slots, and fewer than two later MFMAs) must not be touched.

Matrix instruction (v_mfma_f32_32x32x1_2b_f32, layouts measured by scripts/micro/mfma_k1.hip's bit check): operand lane L belongs to block L >> 5;
D[blk][i][j] = fmaf(A[32 blk + i], B[32 blk + j], C) lands in lane j + 32 ((i >> 2) & 1), register 16 blk + (i & 3) + 4 (i >> 3).
v_permlane32_swap_b32 vdst, vsrc: vdst's upper 32 lanes <-> vsrc's lower 32 lanes."""
import re

import numpy as np

F = np.float32
U = np.uint32
MFMA_WAIT = 18


def _f(x):
    return np.asarray(x, U).view(F)


def _u(x):
    return np.asarray(x, F).view(U)


class Wave:
    def __init__(self, mem, lds_bytes, operands):
        """mem: bytearray-like numpy uint8 (global memory); operands: name -> ("v", np.uint32[64]) | ("s", int) | ("s64", int)."""
        self.mem = mem
        self.lds = np.zeros(lds_bytes, np.uint8)
        self.v = np.zeros((256, 64), U)
        self.s = {}
        self.names = {}
        nxt = 225                              # operands sit above the program's fixed registers (v0..v224; the register form also fixes v240..v255: asserted by the caller)
        for k, (kind, val) in operands.items():
            if kind == "v":
                self.names[k] = ("v", nxt)
                self.v[nxt] = np.asarray(val, U)
                nxt += 1
            else:
                self.names[k] = ("s", k)
                self.s[k] = int(val)
        self.n_operand_vgprs = nxt - 225
        assert nxt <= 256
        self.exec = np.ones(64, bool)
        self.vcc = np.zeros(64, bool)
        self.scc = 0
        self.pending = {}                      # vgpr -> "vm" | "lgkm"
        self.vm_queue = []                     # outstanding vector-memory operations, oldest first: the registers each will write (a store: none)
        self.dma = []                          # LDS byte ranges that an LDS-DMA load still in vm_queue will write: (queue entry, lo, hi)
        self.m0 = 0
        self.mfma = []                         # (set of regs, slot of issue, mfma ordinal)
        self.n_mfma = 0
        self.slot = 0
        self.wrote_at = {}
        self.count = {}

    # ---- operand access ----
    def _vidx(self, tok):
        tok = tok.strip()
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return list(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.fullmatch(r"v(\d+)", tok)
        if m:
            return [int(m.group(1))]
        m = re.fullmatch(r"%\[(\w+)\]", tok)
        if m and self.names[m.group(1)][0] == "v":
            return [self.names[m.group(1)][1]]
        return None

    def _touch(self, regs, what):
        for r in regs:
            assert r not in self.pending, f"{what}: v{r} is still in flight ({self.pending.get(r)}) -- no covering s_waitcnt"
        for (rs, at, idx) in self.mfma:
            if rs & set(regs):
                later = self.n_mfma - idx
                ok = later >= (1 if what.startswith("v_mfma") else 2) or self.slot - at >= MFMA_WAIT
                assert ok, f"{what}: touches an MFMA result {self.slot - at} slots after its issue"

    def _src32(self, tok):
        """A 32-bit source as uint32[64]."""
        tok = tok.strip()
        absv = tok.startswith("|")
        tok = tok.strip("|")
        idx = self._vidx(tok)
        if idx is not None:
            self._touch(idx, "read")
            x = self.v[idx[0]].copy()
        else:
            m = re.fullmatch(r"%\[(\w+)\]", tok)
            if m:
                x = np.full(64, self.s[m.group(1)] & 0xFFFFFFFF, U)
            elif re.fullmatch(r"s(\d+)", tok):
                x = np.full(64, self.s[int(tok[1:])], U)
            elif re.fullmatch(r"-?\d+\.\d+", tok):
                x = np.full(64, _u(F(float(tok))), U)
            elif tok.startswith("0x"):
                x = np.full(64, int(tok, 16), U)
            else:
                x = np.full(64, int(tok) & 0xFFFFFFFF, U)
        if absv:
            x = x & U(0x7FFFFFFF)
        return x

    def _wr(self, tok, val, what="write"):
        idx = self._vidx(tok)
        assert idx is not None and len(idx) == 1, tok
        self._touch(idx, what)
        self.v[idx[0]] = np.where(self.exec, np.asarray(val, U), self.v[idx[0]])
        self.wrote_at[idx[0]] = self.slot

    def _s64(self, tok):
        tok = tok.strip()
        if tok == "vcc":
            return self.vcc.copy()
        if tok == "exec":
            return self.exec.copy()
        m = re.fullmatch(r"%\[(\w+)\]", tok)
        v = self.s[m.group(1)]
        return np.array([(v >> k) & 1 for k in range(64)], bool)

    def _set64(self, tok, mask):
        tok = tok.strip()
        if tok == "vcc":
            self.vcc = mask.copy()
        elif tok == "exec":
            self.exec = mask.copy()
        else:
            m = re.fullmatch(r"%\[(\w+)\]", tok)
            self.s[m.group(1)] = int(sum(1 << k for k in range(64) if mask[k]))

    # ---- execution ----
    def run(self, lines, max_steps=10 ** 8):
        labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
        pc, steps = 0, 0
        while pc < len(lines):
            l = lines[pc]
            pc += 1
            steps += 1
            assert steps < max_steps
            if l.endswith(":"):
                continue
            op, _, rest = l.partition(" ")
            self.count[op] = self.count.get(op, 0) + 1
            off = 0
            rest = re.sub(r"(\s+(nt|sc0|sc1))+$", "", rest)
            m = re.search(r"\s+offset:(\d+)", rest)
            if m:
                off = int(m.group(1))
                rest = rest[:m.start()] + rest[m.end():]
            toks = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", rest)] if rest else []
            if op == "s_nop":
                self.slot += int(toks[0]) + 1
                continue
            self.slot += 1
            if op == "s_waitcnt":
                assert re.fullmatch(r"(vmcnt\(\d+\)|lgkmcnt\(0\)|\s)+", rest), l
                if "lgkmcnt(0)" in rest:
                    self.pending = {r: k for r, k in self.pending.items() if k != "lgkm"}
                m = re.search(r"vmcnt\((\d+)\)", rest)
                if m:                                          # memory operations complete in issue order: all but the newest N are done
                    while len(self.vm_queue) > int(m.group(1)):
                        for r in self.vm_queue.pop(0):
                            if self.pending.get(r) == "vm":
                                del self.pending[r]
                    self.dma = [(e, lo, hi) for (e, lo, hi) in self.dma if any(e is q for q in self.vm_queue)]
                continue
            if op in ("s_branch", "s_cbranch_scc1", "s_cbranch_scc0"):
                if op == "s_branch" or (op == "s_cbranch_scc1") == bool(self.scc):
                    pc = labels[toks[0]]
                continue
            if op in ("s_cmp_gt_u32", "s_cmp_lg_u32", "s_cmp_eq_u32"):
                a, b = [self.s[re.fullmatch(r"%\[(\w+)\]", t).group(1)] if t.startswith("%") else int(t) for t in toks]
                self.scc = int(a > b) if op == "s_cmp_gt_u32" else int(a != b) if op == "s_cmp_lg_u32" else int(a == b)
                continue
            if op == "s_sub_u32":
                name = re.fullmatch(r"%\[(\w+)\]", toks[0]).group(1)
                a = self.s[re.fullmatch(r"%\[(\w+)\]", toks[1]).group(1)]
                self.s[name] = (a - int(toks[2])) & 0xFFFFFFFF
                continue
            if op == "s_mov_b64":
                self._set64(toks[0], self._s64(toks[1]))
                continue
            if op == "s_and_b64":
                self._set64(toks[0], self._s64(toks[1]) & self._s64(toks[2]))
                continue
            if op == "s_and_saveexec_b64":
                old = self.exec.copy()
                self.exec = self._s64(toks[1]) & old
                self._set64(toks[0], old)
                continue
            if op.startswith("s_load_dwordx"):
                n = int(op[len("s_load_dwordx"):])
                m = re.fullmatch(r"s\[(\d+):(\d+)\]", toks[0])
                base = self.s[re.fullmatch(r"%\[(\w+)\]", toks[1]).group(1)] + int(toks[2], 16)
                assert int(m.group(2)) - int(m.group(1)) + 1 == n
                words = self.mem[base:base + 4 * n].view(U)
                for k in range(n):
                    self.s[int(m.group(1)) + k] = int(words[k])
                continue
            if op.startswith("global_load_dwordx"):
                n = int(op[len("global_load_dwordx"):])
                dst = self._vidx(toks[0])
                assert len(dst) == n and dst[0] % 2 == 0
                addr = self._src32(toks[1]).astype(np.int64) + self.s[re.fullmatch(r"%\[(\w+)\]", toks[2]).group(1)] + off
                self._touch(dst, op)
                for lane in np.nonzero(self.exec)[0]:
                    a = int(addr[lane])
                    assert a % (4 * min(n, 4)) == 0 and 0 <= a and a + 4 * n <= len(self.mem), (l, lane, a)
                    w = self.mem[a:a + 4 * n].view(U)
                    for k in range(n):
                        self.v[dst[k], lane] = w[k]
                for r in dst:
                    self.pending[r] = "vm"
                self.vm_queue.append(list(dst))
                continue
            if op == "global_store_dwordx4":
                data = self._vidx(toks[1])
                self._touch(data, op)
                addr = self._src32(toks[0]).astype(np.int64) + self.s[re.fullmatch(r"%\[(\w+)\]", toks[2]).group(1)] + off
                for lane in np.nonzero(self.exec)[0]:
                    a = int(addr[lane])
                    assert a % 16 == 0 and a + 16 <= len(self.mem)
                    self.mem[a:a + 16] = np.array([self.v[r, lane] for r in data], U).view(np.uint8)
                self.vm_queue.append([])
                continue
            if op == "s_add_u32" and toks[0] == "m0":
                self.m0 = (self.s[re.fullmatch(r"%\[(\w+)\]", toks[1]).group(1)] + int(toks[2])) & 0xFFFFFFFF
                continue
            if op == "s_mov_b32":
                if toks[0] == "m0":
                    self.m0 = self.s[re.fullmatch(r"%\[(\w+)\]", toks[1]).group(1)]
                else:
                    assert toks[1] == "m0"
                    self.s[re.fullmatch(r"%\[(\w+)\]", toks[0]).group(1)] = self.m0
                continue
            if op == "global_load_lds_dwordx4":                # 16 bytes per lane from its own address to LDS[M0 + 16 lane]: no register written
                addr = self._src32(toks[0]).astype(np.int64) + self.s[re.fullmatch(r"%\[(\w+)\]", toks[1]).group(1)] + off
                entry = []
                for lane in np.nonzero(self.exec)[0]:
                    a, d = int(addr[lane]), self.m0 + off + 16 * int(lane)
                    assert a % 16 == 0 and 0 <= a and a + 16 <= len(self.mem) and d + 16 <= len(self.lds), (l, lane, a, d)
                    self.lds[d:d + 16] = self.mem[a:a + 16]
                self.vm_queue.append(entry)
                self.dma.append((entry, self.m0 + off, self.m0 + off + 1024))
                continue
            if op in ("ds_read_b128", "ds_write_b128"):
                rd = op == "ds_read_b128"
                regs = self._vidx(toks[0] if rd else toks[1])
                assert len(regs) == 4 and regs[0] % 2 == 0
                addr = self._src32(toks[1] if rd else toks[0]).astype(np.int64) + off
                self._touch(regs, op)
                live = [(lo, hi) for (e, lo, hi) in self.dma if any(e is q for q in self.vm_queue)]
                for lane in np.nonzero(self.exec)[0]:
                    a = int(addr[lane])
                    assert a % 16 == 0 and a + 16 <= len(self.lds), (l, lane, a)
                    assert not any(lo < a + 16 and a < hi for lo, hi in live), f"{l}: LDS bytes {a}.. are the target of an LDS-DMA load still in flight (no covering vmcnt)"
                    if rd:
                        w = self.lds[a:a + 16].view(U)
                        for k in range(4):
                            self.v[regs[k], lane] = w[k]
                    else:
                        self.lds[a:a + 16] = np.array([self.v[r, lane] for r in regs], U).view(np.uint8)
                if rd:
                    for r in regs:
                        self.pending[r] = "lgkm"
                continue
            if op == "v_mfma_f32_32x32x1_2b_f32":
                dst = self._vidx(toks[0])
                assert len(dst) == 32 and toks[3] == "0"
                ia, ib = self._vidx(toks[1]), self._vidx(toks[2])
                for r in ia + ib:
                    assert self.slot - self.wrote_at.get(r, -99) > 2, f"{l}: operand v{r} written {self.slot - self.wrote_at[r]} slots ago"
                self._touch(dst, "v_mfma write")
                a, b = _f(self._src32(toks[1])), _f(self._src32(toks[2]))
                res = np.zeros((32, 64), F)
                with np.errstate(all="ignore"):
                    for blk in range(2):
                        prod = (a[32 * blk:32 * blk + 32, None] * b[None, 32 * blk:32 * blk + 32]).astype(F)     # [i][j]
                        zero = (a[32 * blk:32 * blk + 32, None] == 0) | (b[None, 32 * blk:32 * blk + 32] == 0)
                        prod = np.where(zero & np.isfinite(prod), F(0.0), prod)                                   # fmaf(a, b, +0): an exact zero is +0
                        for i in range(32):
                            res[16 * blk + (i & 3) + 4 * (i >> 3), 32 * ((i >> 2) & 1):32 * ((i >> 2) & 1) + 32] = prod[i]
                for k in range(32):
                    self.v[dst[k]] = _u(res[k])
                self.n_mfma += 1
                self.mfma = [(rs, at, idx) for (rs, at, idx) in self.mfma if self.n_mfma - idx < 3]
                self.mfma.append((set(dst), self.slot, self.n_mfma))
                continue
            if op == "v_permlane32_swap_b32":
                ia, ib = self._vidx(toks[0])[0], self._vidx(toks[1])[0]
                for r in (ia, ib):
                    assert self.slot - self.wrote_at.get(r, -99) > 2, f"{l}: v{r} written {self.slot - self.wrote_at[r]} slots ago"
                self._touch([ia, ib], op)
                hi = self.v[ia, 32:].copy()
                self.v[ia, 32:] = self.v[ib, :32]
                self.v[ib, :32] = hi
                continue
            if op in ("v_cmp_eq_u32", "v_cmp_lt_i32"):
                a, b = self._src32(toks[1]), self._src32(toks[2])
                r = (a == b) if op == "v_cmp_eq_u32" else (a.view(np.int32) < b.view(np.int32))
                assert toks[0] == "vcc"
                self.vcc = np.where(self.exec, r, False)
                continue
            if op in ("v_cndmask_b32", "v_cndmask_b32_e64"):
                a, b = self._src32(toks[1]), self._src32(toks[2])
                assert toks[3] == "vcc"
                self._wr(toks[0], np.where(self.vcc, b, a))
                continue
            if op == "v_mov_b32":
                self._wr(toks[0], self._src32(toks[1]))
                continue
            if op in ("v_xor_b32", "v_add_u32", "v_and_b32", "v_lshrrev_b32"):
                a, b = self._src32(toks[1]), self._src32(toks[2])
                r = {"v_xor_b32": lambda: a ^ b, "v_add_u32": lambda: (a + b).astype(U), "v_and_b32": lambda: a & b, "v_lshrrev_b32": lambda: b >> (a & U(31))}[op]()
                self._wr(toks[0], r)
                continue
            if op in ("v_mul_f32", "v_add_f32", "v_sub_f32", "v_max_f32"):
                a, b = _f(self._src32(toks[1])), _f(self._src32(toks[2]))
                with np.errstate(all="ignore"):
                    r = {"v_mul_f32": lambda: a * b, "v_add_f32": lambda: a + b, "v_sub_f32": lambda: a - b, "v_max_f32": lambda: np.maximum(a, b)}[op]()
                self._wr(toks[0], _u(r.astype(F)))
                continue
            if op == "v_pk_add_f32":                            # two one-lane additions on a register pair, each rounded on its own (no op_sel / neg modifiers here)
                lo = [int(re.fullmatch(r"v\[(\d+):(\d+)\]", t).group(1)) for t in toks]
                assert all(x % 2 == 0 for x in lo), l
                res = []
                for h in (0, 1):
                    a, b = _f(self._src32(f"v{lo[1] + h}")), _f(self._src32(f"v{lo[2] + h}"))
                    with np.errstate(all="ignore"):
                        res.append(_u((a + b).astype(F)))
                for h in (0, 1):
                    self._wr(f"v{lo[0] + h}", res[h])
                continue
            raise AssertionError("wave emulator: unknown instruction: " + l)
        return self
