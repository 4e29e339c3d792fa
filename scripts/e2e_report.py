#!/usr/bin/env python3
"""Measure how the default-mode candidate lists differ from the reference chain's, end to end from IQ
(tests/e2e_candidates.py); writes gpurun_out/e2e_candidates.json (copied to profiles/ per round)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cwsl_digi_amd as P                      # noqa: E402
from oracle import oracle as O                 # noqa: E402  (checker only)
import e2e_candidates as E                     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
O.build(ref=False)
out = {}
for mode, seed0, smin in (("FT8", 4100, 1.5), ("FT4", 4200, 1.2)):
    slots = E.make_slots(O, mode, n, seed0)
    ref = E.run_oracle(O, mode, slots)
    for exact in (False, True):
        with P.Context(0) as ctx:
            gpu = E.run_gpu(ctx, mode, slots, exact)
        out[f"{mode}_{'exact (the default of a new context)' if exact else 'fast'}"] = E.compare(mode, gpu, ref, smin)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "e2e_candidates.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
