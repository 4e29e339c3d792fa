#!/usr/bin/env python3
"""Randomised soak of the GPU path against the oracle (run on the GPU box): random rates, block lengths, tunings,
modes, push granularities and signal mixes.  Exact mode must be bit-identical (float + int16), the default mode within
1e-5 of frame peak (int16 within 1 LSB), FT8/FT4 candidate lists and FT4 refined records identical.  One JSON line.

    python scripts/gpu_soak.py --seconds 300 [--seed 1]
"""
import argparse, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch  # noqa: F401
    import cwsl_digi_amd as P
    from oracle import oracle as O
    from ft8_signal import ft8_iq, ft4_iq

    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    st = dict(iterations=0, retunes=0, exact_frames=0, fast_frames=0, ft8_lists=0, ft4_lists=0, ft4_records=0, candidates=0,
              worst_fast_rel=0.0, failures=[])
    period = {"FT8": 15.0, "FT4": 7.5, "JT65": 6.0}
    while time.time() < t_end and not st["failures"]:
        fs = int(rng.choice([48000, 96000, 192000]))
        D = fs // 12000
        blk = int(4 * D * rng.integers(1, 40))
        mode = str(rng.choice(["FT8", "FT4", "FT8", "FT4", "JT65"]))
        half = fs // 2
        f = int(rng.integers(-half, half - 6000))
        exact = bool(rng.integers(0, 2))
        slot_s = period[mode] * float(rng.choice([1.0, 1.0, 0.37]))
        n = int(slot_s * fs) // blk * blk
        if n == 0:
            continue
        iq = O.synth_iq(int(rng.integers(1, 1 << 30)), n, fs, tones_hz=[f + float(rng.uniform(300, 2700))], amp=float(rng.uniform(0, 8000)))
        iq = iq * np.float32(rng.uniform(0.01, 1.0))
        for _ in range(int(rng.integers(0, 4))):
            a_hz, t0 = float(rng.uniform(300, 2600)), float(rng.uniform(0.0, max(0.1, slot_s - 5.0)))
            gen = ft4_iq if mode == "FT4" else ft8_iq
            iq = iq + gen(fs, n, f, a_hz, t0, float(rng.uniform(500, 4000)), rng)
        iq = iq.astype(np.complex64)
        max_cand, f_hi = int(rng.choice([50, 200])), int(rng.choice([2500, 3000, 3600]))
        tag = dict(it=st["iterations"], fs=fs, blk=blk, mode=mode, f=f, exact=exact, n=n, max_cand=max_cand, f_hi=f_hi)
        ctx = P.Context(0)
        try:
            ctx.set_exact(exact)
            ctx.enable_sync(True, 1.5, max_cand, 200, f_hi)
            rx = ctx.receiver_open(fs, blk, 0)
            ch = ctx.channel_open(rx, f, mode)
            for _ in range(int(rng.integers(0, 3))):            # neighbours sharing the receiver's IQ (must not matter)
                ctx.channel_open(rx, int(rng.integers(-half, half - 6000)), str(rng.choice(["FT8", "FT4"])))
            if mode == "FT8" and rng.random() < 0.2:            # enough FT8 channels for the per-channel form of the sync stage (>= 2 per CU)
                for _ in range(520):
                    ctx.channel_open(rx, int(rng.integers(-half, half - 6000)), "FT8")
                st["many_channel_iterations"] = st.get("many_channel_iterations", 0) + 1
            oc = O.Channel(mode, fs, blk, f)
            ctx.slot_boundary(mode, 10); oc.boundary(10)
            order = str(rng.choice(["sync", "sync", "freq"]))   # round 6: cwslg_set_candidate_order
            ctx.set_candidate_order(order)
            # round 6: one to three slots per context, of different lengths -- a short one, a late boundary (the 20 s frame's tail in use), an ordinary one --
            # so that the fused finalise's tail bookkeeping (what the int16 buffer held before) and the epoch carried by every result are exercised
            n_full = n
            for slot_k in range(int(rng.choice([1, 1, 2, 3]))):
              epoch = 25 + 15 * slot_k
              if slot_k:
                  n = max(blk, int(n_full * float(rng.choice([0.3, 1.0, 1.12]))) // blk * blk)
                  iq = np.concatenate([iq, iq])[:n] if n > len(iq) else iq[:n]
              retune_at = int(rng.integers(1, max(2, n // blk))) * blk if (rng.random() < 0.25 and slot_k == 0) else -1
              f2 = int(rng.integers(-half, half - 6000))
              pos = 0
              while pos < n:                          # ragged multi-block pushes; sometimes SSBD::Tune in mid-slot
                  m = min(n - pos, blk * int(rng.integers(1, 60)))
                  if 0 <= retune_at - pos < m and retune_at > pos:
                      m = retune_at - pos
                  if pos == retune_at:                # SSBD::Tune with reset = true or false (SSBD.hpp:97), sometimes flipping the sideband
                      keep, usb2 = bool(rng.random() < 0.5), bool(rng.random() < 0.8)
                      try:
                          oc.tune(f2, usb2, reset=not keep)
                      except ValueError:              # out of band for this sideband: both sides refuse and keep the old tuning
                          try:
                              ctx.channel_tune(ch, f2, usb2, reset=not keep)
                              st["failures"].append(dict(tag, what="retune accepted that the reference refuses"))
                          except P.CwslGpuError:
                              pass
                      else:
                          ctx.channel_tune(ch, f2, usb2, reset=not keep)
                      st["retunes"] += 1; st["retunes_keep"] = st.get("retunes_keep", 0) + int(keep)
                  ctx.push_iq(rx, iq[pos:pos + m]); oc.push_many(iq[pos:pos + m]); pos += m
              ctx.slot_boundary(mode, epoch)
              ref = oc.boundary(epoch, want_f32=True)
              a, nv = ctx.fetch_audio_f32(ch)
              g = ctx.fetch_frame(ch)
              if exact:
                  if not (np.array_equal(a.view(np.uint32), ref["f32"].view(np.uint32)) and np.array_equal(g["i16"], ref["i16"])):
                      st["failures"].append(dict(tag, what="exact frame differs"))
                  st["exact_frames"] += 1
              else:
                  peak = float(np.abs(ref["f32"]).max()) or 1.0
                  rel = float(np.abs(a.astype(np.float64) - ref["f32"]).max()) / peak
                  st["worst_fast_rel"] = max(st["worst_fast_rel"], rel)
                  if rel > 1e-5 or int(np.abs(g["i16"].astype(np.int32) - ref["i16"]).max()) > 1:
                      st["failures"].append(dict(tag, what="fast frame out of tolerance", rel=rel))
                  st["fast_frames"] += 1
              # the sync stage works on the frame the GPU produced (g), whatever the mode's arithmetic
              if mode == "FT8":
                  got, t_list = ctx.fetch_candidates(ch, with_epoch=True)
                  sl = ctx.fetch_slot(ch, max_list=max_cand)
                  if t_list != g["t_start"] or sl["t_start"] != g["t_start"] or sl["list_kind"] != "FT8" or sl["list"] != got or not np.array_equal(sl["i16"], g["i16"]):
                      st["failures"].append(dict(tag, what="fetch_slot / epoch inconsistent", slot=slot_k))
                  want = O.ft8_sync(g["i16"], 200, f_hi, 1.5, max_cand, order=order)
                  if [tuple(x) for x in got] != [tuple(x) for x in want]:
                      st["failures"].append(dict(tag, what="ft8 candidate list differs", n_got=len(got), n_want=len(want)))
                  st["ft8_lists"] += 1; st["candidates"] += len(want)
              elif mode == "FT4":
                  got = ctx.fetch_candidates(ch)
                  want = O.ft4_candidates(g["i16"], 200.0, float(f_hi), 1.2, max_cand, order=order)
                  if [tuple(x) for x in got] != [tuple(x) for x in want]:
                      st["failures"].append(dict(tag, what="ft4 candidate list differs", n_got=len(got), n_want=len(want)))
                  got4, want4 = ctx.fetch_ft4_sync(ch), O.ft4_sync_all(g["i16"], want)
                  if got4 != want4:
                      st["failures"].append(dict(tag, what="ft4 refined records differ", n_got=len(got4 or []), n_want=len(want4)))
                  st["ft4_lists"] += 1; st["candidates"] += len(want); st["ft4_records"] += len(want4)
        finally:
            ctx.close()
        st["iterations"] += 1
    print(json.dumps(st))
    sys.exit(1 if st["failures"] else 0)


if __name__ == "__main__":
    main()
