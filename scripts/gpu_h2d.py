"""PCIe-inclusive rate: host buffers through cwslg_push_iq (pageable numpy -> this receiver's pinned staging -> H2D on the copy
stream -> demod), with 1, 2, 4, 8 host threads each feeding its own receivers -- the reference's shape, one thread per Receiver
(Receiver.hpp:167).  ctypes releases the GIL during the call, so the threads really copy in parallel."""
import json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cwsl_digi_amd as P

S, BLK, NB = 64, 2048, 940            # 64 receivers x 10 s
rng = np.random.default_rng(0)
blk = (rng.standard_normal(BLK * 64) + 1j * rng.standard_normal(BLK * 64)).astype(np.complex64) * 1000
out = {}
for threads in (1, 2, 4, 8):
    with P.Context(0) as ctx:
        rxs = [ctx.receiver_open(192000, BLK, 0) for _ in range(S)]
        for k, rx in enumerate(rxs):
            ctx.channel_open(rx, -80000 + 2500 * k, "FT8")
        ctx.slot_boundary("FT8", 1)
        for rx in rxs:
            ctx.push_iq(rx, blk[:BLK])          # allocate every receiver's staging outside the timed region
        ctx.synchronize()

        def feed(mine):
            for it in range(NB // 64):
                for rx in mine:
                    ctx.push_iq(rx, blk)        # 64 blocks per call

        th = [threading.Thread(target=feed, args=(rxs[t::threads],)) for t in range(threads)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        ctx.slot_boundary("FT8", 2)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        n = S * (NB // 64) * 64 * BLK
        assert ctx.stats()["demod_samples"] == n + S * BLK
        out[threads] = {"msamples_s": n / dt / 1e6, "gb_s": n * 8 / dt / 1e9, "realtime_192k_streams": n / dt / 192000}
        print("host-buffer path, %d pusher thread(s): %.1f M samples/s = %.2f GB/s over PCIe, %.0f real-time 192 kHz streams"
              % (threads, n / dt / 1e6, n * 8 / dt / 1e9, n / dt / 192000))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/h2d.json", "w"), indent=1)
