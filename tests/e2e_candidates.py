"""End-to-end candidate-list comparison, from IQ (shared by tests/test_gpu_e2e_candidates.py and
scripts/e2e_report.py).

Reference chain (CPU):  IQ -> oracle.Channel (SSBD + framing + prepareAudio + int16, pinned against the compiled
reference headers) -> int16 frame -> ft8_sync / ft4_candidates / ft4_sync_all (the sync restatement).
Product chain (GPU):    IQ -> cwslg_push_iq -> cwslg_slot_boundary -> cwslg_fetch_candidates / _fetch_ft4_sync.
Unlike tests/test_gpu_sync.py, the restatement is NOT fed the GPU's own int16 frame: the two chains only share the IQ
(Instance.cpp:230-245 -> frame -> DecoderPool.hpp:634-640).
"""
import numpy as np

from ft8_signal import ft8_iq, ft4_iq

FS, BLK = 192000, 2048


def make_slots(oracle, mode, n_slots, seed0):
    """n_slots private-stream slots with 3-6 bursts each over noise; returns [(f, iq)]."""
    n = (2880000 if mode == "FT8" else 1440000) // BLK * BLK
    out = []
    for s in range(n_slots):
        rng = np.random.default_rng(seed0 + s)
        f = int(-85000 + (s * 23917) % 170000)
        iq = oracle.synth_iq(seed0 + s, n, FS)                      # noise, sigma ~ 1000
        nb = 3 + s % 4
        for k in range(nb):
            a_hz = float(rng.uniform(350.0, 2700.0))
            amp = float(rng.uniform(400.0, 3000.0))                # from near the threshold to strong
            if mode == "FT8":
                iq = iq + ft8_iq(FS, n, f, a_hz, float(rng.uniform(0.0, 1.8)), amp, rng)
            else:
                iq = iq + ft4_iq(FS, n, f, a_hz, float(rng.uniform(0.0, 1.5)), amp, rng)
        out.append((f, iq.astype(np.complex64)))
    return out


def run_gpu(ctx, mode, slots, exact, maxcand=200):
    if exact is not None:                 # None: leave the context as cwslg_create() made it
        ctx.set_exact(exact)
    ctx.enable_sync(True, 1.5, maxcand, 200, 3000)
    res = []
    for f, iq in slots:
        rx = ctx.receiver_open(FS, BLK, 0)
        ch = ctx.channel_open(rx, f, mode)
        ctx.slot_boundary(mode, 1)
        for k in range(0, len(iq), 128 * BLK):
            ctx.push_iq(rx, iq[k:k + 128 * BLK])
        ctx.slot_boundary(mode, 16)
        fr = ctx.fetch_frame(ch)["i16"].copy()
        cands = ctx.fetch_candidates(ch, maxcand)
        rec = ctx.fetch_ft4_sync(ch) if mode == "FT4" else None
        res.append((fr, cands, rec))
        ctx.receiver_close(rx)
    return res


def run_oracle(oracle, mode, slots, maxcand=200):
    res = []
    for f, iq in slots:
        oc = oracle.Channel(mode, FS, BLK, f)
        assert oc.boundary(1) is None
        oc.push_many(iq)
        fr = oc.boundary(16)["i16"]
        if mode == "FT8":
            cands = oracle.ft8_sync(fr, 200, 3000, 1.5, maxcand)
            rec = None
        else:
            cands = oracle.ft4_candidates(fr, 200.0, 3000.0, 1.2, maxcand)
            rec = oracle.ft4_sync_all(fr, cands)
        res.append((fr, cands, rec))
    return res


def compare(mode, gpu, ref, syncmin):
    """Per slot: are the lists identical bit for bit?  If not: which (bin, lag) keys differ and by how much the shared
    ones' sync values move.  A key present on one side only is 'marginal' when its sync is within 1e-3 of the threshold
    or of the last rank kept."""
    rep = dict(mode=mode, slots=len(gpu), identical_lists=0, int16_mismatches=[], n_cands=[], worst_rel_sync=0.0,
               only_one_side=0, only_one_side_not_marginal=0, order_changes=0, ft4_records_identical=0,
               ft4_worst_f1_hz=0.0, ft4_worst_dt_s=0.0, ft4_worst_rel_sync=0.0, ft4_records_unmatched=0)
    for (gfr, gc, grec), (rfr, rc, rrec) in zip(gpu, ref):
        rep["int16_mismatches"].append(int((gfr != rfr).sum()))
        rep["n_cands"].append(len(rc))
        key = (lambda c: (c[0], c[1])) if mode == "FT8" else (lambda c: (c[0],))
        bits = lambda c: (c[0], c[1], np.float32(c[2]).view(np.uint32), np.float32(c[3]).view(np.uint32))
        if [bits(c) for c in gc] == [bits(c) for c in rc]:
            rep["identical_lists"] += 1
        gd, rd = {key(c): c for c in gc}, {key(c): c for c in rc}
        for k in set(gd) & set(rd):
            rel = abs(float(gd[k][2]) - float(rd[k][2])) / max(abs(float(rd[k][2])), 1e-30)
            rep["worst_rel_sync"] = max(rep["worst_rel_sync"], rel)
        floor = min([float(c[2]) for c in rc] + [float(c[2]) for c in gc] + [1e30])
        for k in set(gd) ^ set(rd):
            c = gd.get(k) or rd.get(k)
            rep["only_one_side"] += 1
            if abs(float(c[2]) - syncmin) > 1e-3 * syncmin and abs(float(c[2]) - floor) > 1e-3 * floor:
                rep["only_one_side_not_marginal"] += 1
        shared_g = [key(c) for c in gc if key(c) in rd]
        shared_r = [key(c) for c in rc if key(c) in gd]
        rep["order_changes"] += int(shared_g != shared_r)
        if mode == "FT4":
            gk = {(r["cand"], r["seg"]): r for r in grec} if [key(c) for c in gc] == [key(c) for c in rc] else {}
            rk = {(r["cand"], r["seg"]): r for r in rrec}
            same = bool(gk) and set(gk) == set(rk)
            for k in set(gk) & set(rk):
                a, b = gk[k], rk[k]
                if (a["ibest"], a["idf"]) != (b["ibest"], b["idf"]) or \
                        np.float32(a["sync"]).view(np.uint32) != np.float32(b["sync"]).view(np.uint32):
                    same = False
                rep["ft4_worst_f1_hz"] = max(rep["ft4_worst_f1_hz"], abs(a["f1_hz"] - b["f1_hz"]))
                rep["ft4_worst_dt_s"] = max(rep["ft4_worst_dt_s"], abs(a["dt_s"] - b["dt_s"]))
                rep["ft4_worst_rel_sync"] = max(rep["ft4_worst_rel_sync"], abs(a["sync"] - b["sync"]) / max(abs(b["sync"]), 1e-30))
            rep["ft4_records_unmatched"] += len(set(gk) ^ set(rk)) if gk else len(rk)
            rep["ft4_records_identical"] += int(same)
    return rep
