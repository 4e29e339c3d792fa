#!/usr/bin/env python3
"""bench.py -- throughput of the CWSL_DIGI hot path on MI355X.

A "step" is one pass of the whole hot path over one batch of synthetic input: every FT8 slot
(channel) owned by this rank demodulates one complete 15 s slot (2 880 000 complex samples at
192 kHz, private stream per slot), the slot boundary fires, every frame is peak-normalised and
rounded to int16 and the FT8 sync stage runs on every frame.  Inputs are resident
in HBM before the timed region (each receiver's ring holds a full slot, filled by the device-side
synthetic source; the timed region re-commits it lap after lap without copying).

Headline = the PRODUCT DEFAULT: the mode cwslg_create() hands out -- reference-order float32 arithmetic ("exact"), frames and candidate
lists bit-identical to the reference chain's.  On one GPU the throughput mode (cwslg_set_exact(ctx, 0): fused polyphase form, audio within
1e-5 of frame peak) is measured right after it on the same slots and reported as the "fast" record of the same JSON line.  --fast swaps
the two; --primary-only skips the second record.

Workload.  The north-star single-GPU workload -- 4096 FT8 slots resident on ONE MI355X (94 GB of IQ + 20 GB of frames,
checkpoints and spectra) -- PER GPU at every N: the per-GPU work is fixed as N grows ("weak"; N = 8 carries 32 768 slots).
--slots overrides it (--slots 512 at N = 8 is BASELINE configs[3], the north star's 4096 slots sharded over eight GPUs).

One process per GPU.  `python bench.py --gpus N` with N > 1 and no RANK in the environment LAUNCHES the N ranks itself: the parent --
before it imports torch or touches a GPU -- starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port P bench.py <same arguments>` as a child process and exits with its status (never an exec from a process that has initialised
the GPU).  Launched by torch.distributed.run directly (RANK / WORLD_SIZE set) it is a rank; WORLD_SIZE != --gpus is an error.
Slots shard across ranks (slot s of rank r is global slot r*S+s) with no data-path collective; the only collective is one
32-byte-per-rank all-gather on RCCL at every slot boundary (the north_star's "barrier on the mode's slot
boundary"), issued from INSIDE cwslg_slot_boundary_end by the library's own communicator (cwslg_rccl_init;
--rendezvous torch hands torch.distributed in through cwslg_set_boundary_rendezvous instead), one boundary late so that it
overlaps the next slot's demod launch.  value = all ranks' samples / max-over-ranks time.  --dry-run exercises the launcher and the
process group on CPU (gloo) without a GPU context.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 192000
IQ_LEN = 2048
SLOT_SAMPLES = 2880000                    # 15 s FT8 slot at 192 kHz
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_SAMPLE_DEMOD = 8.0 + 4.0 / 16   # demod kernel: 8 B IQ read + 0.25 B float audio write
BYTES_PER_SAMPLE_PATH = 8.625             # + finalise: 0.25 B read + 0.125 B int16 write
VALU_PEAK_TFLOPS = 157.3                  # MI355X_MICROARCH.md: FP32 vector peak (spec), an FMA counted as 2
# FT8 sync stage, floating-point operations of the "spec v3" arithmetic (oracle/sync_oracle.c; an FMA = 2), per 3840-point symbol transform:
#   scale by 1/300                                  3 840
#   stage 1 (prime-factor 3 x 5 on the 8 live inputs): 128 columns x (five-point sums 122 + five three-point DFTs 100 + 14 twiddle products 84) = 128 x 306 = 39 168
#   stage 2: 15 rows x 64 butterflies x (len 2, 4: 4 + 4; len 8: half plain 4, half three-fmaf 12; len 16..128: 4 x 12)                        = 61 440
#   real-input unpack + |X|^2 of the stored bins:   17 per bin
# (spec v2, rounds 2-3: 3 840 + 69 888 + 67 200 + 20 per bin = 160 768 at 992 bins; v3: 121 312 -- a quarter fewer operations for the same transform,
# so the stage's TFLOP/s figure FALLS while its time does: compare times across rounds, not fractions of the FP32 peak)
# and per searched bin of the Costas stage: 7-tone sums 6 x 378 adds, 125 lags x (42 adds + 30 for the two sync ratios) = 11 268
def sync_flops_per_slot(nbins, n_search_bins):
    per_transform = 3840 + 39168 + 61440 + 17 * nbins
    return 372 * per_transform + n_search_bins * 11268, per_transform


def host_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(-(-int(q) // int(per)))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, -(-q // per)))
        except Exception:
            pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown CPU"


def slot_freq(gs):
    """Tuning offset of global slot gs: spread over the legal band (|F|<=96k, |F+6k|<=96k)."""
    return -90000 + (gs * 4373) % 176000


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks (one per GPU); N > 1 without RANK in the environment launches them through torch.distributed.run")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--slots", type=int, default=0, help="FT8 slots per GPU (default: 4096 = the north-star workload, at every N; 512 at N = 8 is configs[3])")
    ap.add_argument("--sync", type=int, default=1, help="run the FT8 sync stage (symbol spectra + Costas search) at every boundary")
    ap.add_argument("--channels-per-rx", type=int, default=1,
                    help="1 = private IQ stream per slot (BASELINE configs); C>1 = the reference's topology, C decoders share one receiver's IQ "
                         "(CWSL: up to 32 receivers, CWSL_Utils.hpp:9; 128 per receiver makes the north star's 4096 channels)")
    ap.add_argument("--fast", action="store_true", help="headline record in the throughput mode (cwslg_set_exact(ctx, 0)); the product default becomes the second record")
    ap.add_argument("--exact", action="store_true", help="(the default since round 4) headline record in the product's default mode: reference-order arithmetic, bit-identical")
    ap.add_argument("--primary-only", action="store_true", help="skip the second record (the other arithmetic mode)")
    ap.add_argument("--fast-only", action="store_true", help="= --fast --primary-only")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (CPU tensors; for checking the N>1 path on a 1-GPU box)")
    ap.add_argument("--same-device", action="store_true", help="testing only: every rank uses GPU 0")
    ap.add_argument("--rendezvous", default="auto", choices=["auto", "builtin", "torch"],
                    help="N > 1: the slot-boundary rendezvous inside cwslg_slot_boundary_end -- builtin = the library's own RCCL all-gather "
                         "(cwslg_rccl_init; no Python inside the boundary), torch = a torch.distributed all-reduce handed in as a callback; "
                         "auto = builtin on the nccl backend with one GPU per rank, torch otherwise (gloo, --same-device: RCCL refuses two ranks on one GPU)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / process-group check without a GPU: every rank joins a gloo group, rank 0 prints the line's skeleton")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline wall time")
    ap.add_argument("--verify", type=int, default=8, help="slots (spread over the whole range) checked against the oracle after the timed region")
    ap.add_argument("--no-spans", action="store_true", help="diagnostic: no HIP timing events around the kernels (roofline fields are then empty)")
    ap.add_argument("--host-timing", action="store_true", help="print the host time spent inside each asynchronous call of a step (stderr)")
    args = ap.parse_args(argv)
    if args.fast_only:
        args.fast, args.primary_only = True, True
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    return args


def launch_ranks(args, argv):
    """The parent of an N > 1 run: nothing GPU- or torch-related has been imported.  Starts torch.distributed.run as a CHILD process (one rank
    per GPU, rendezvous on 127.0.0.1) and returns its exit status."""
    import socket
    import subprocess
    with socket.socket() as sk:                       # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("bench.py: launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, rank, world):
    """No GPU: the ranks join a gloo group on 127.0.0.1, agree on the world size and rank 0 prints the line's skeleton."""
    import torch
    import torch.distributed as dist
    from cwsl_digi_amd import shard
    S = args.slots if args.slots > 0 else 4096
    seen = world
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.tensor([1], dtype=torch.int64)
        dist.all_reduce(t)
        seen = int(t.item())
        every = [None] * world
        dist.all_gather_object(every, list(shard.slots_of_rank(S * world, rank, world))[:1])
        assert [e[0] for e in every] == [r * S for r in range(world)], every
        dist.barrier()
    assert seen == world == args.gpus
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "dry_run": True, "ranks_seen": seen, "scaling": "weak", "config": {"workload": f"dry run: {S} FT8 slots/GPU x {world} ranks, no GPU work"}}))
    if world > 1:
        dist.destroy_process_group()


METRIC = "IQ Msamples/s demod+sync per GPU; concurrent FT8 slots at real-time; % HBM roofline"
MODE_TEXT = {True: "exact: the product default (cwslg_create); reference-order un-fused float32, frames and candidate lists bit-identical to the reference chain",
             False: "fast: fused polyphase arithmetic (cwslg_set_exact(ctx, 0)), float audio within 1e-5 of frame peak"}


def main():
    args = parse_args()
    env_rank, env_world = os.environ.get("RANK"), os.environ.get("WORLD_SIZE")
    if env_rank is None and env_world is None:
        if args.gpus > 1:                               # the parent: launch the ranks, exit with their status -- before torch / HIP is touched
            sys.exit(launch_ranks(args, sys.argv[1:]))
    rank = int(env_rank or "0")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if world != args.gpus:
        print(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: launch with --nproc-per-node {args.gpus} (or drop RANK/WORLD_SIZE and let "
              f"bench.py --gpus {args.gpus} launch the ranks itself)", file=sys.stderr)
        sys.exit(2)
    if args.dry_run:
        return dry_run(args, rank, world)

    import torch                          # first: its libamdhip64 must be the process's HIP runtime
    import torch.distributed as dist
    import numpy as np
    import cwsl_digi_amd as P
    from cwsl_digi_amd import shard

    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    dev = torch.device("cuda", local_rank) if args.dist_backend == "nccl" else torch.device("cpu")

    S = args.slots if args.slots > 0 else 4096
    ctx = P.Context(local_rank)
    rendezvous = None
    builtin_failed = builtin_failed_kind = None
    if world > 1:
        rendezvous = args.rendezvous
        if rendezvous == "auto":
            rendezvous = "builtin" if (args.dist_backend == "nccl" and not args.same_device) else "torch"
        if rendezvous == "builtin":
            # The library's own RCCL communicator, brought up in three steps that EVERY rank agrees on (over the process group torch already has)
            # before the next one starts -- a rank must never sit alone inside a collective:
            #   load    librccl opens on every rank (each makes an ncclUniqueId; rank 0's is the one that is used)
            #   init    cwslg_rccl_init (ncclCommInitRank) returns on every rank
            #   gather  one boundary of a mode group without channels: the 32-byte all-gather itself runs on the new communicator
            # The first step that fails anywhere names the reason class (multi_gpu.builtin_failed_kind) and the run goes on with the
            # torch.distributed callback rendezvous.
            def agree(err):
                ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 1:
                    return None
                why = [None] * world
                dist.all_gather_object(why, ("rank %d: %s" % (rank, err)) if err else None)     # (the exception text is the library's cwslg_last_error)
                return "; ".join(w for w in why if w)
            err, uid = None, b""
            try:
                uid = P.rccl_unique_id()
            except Exception as e:                  # noqa: BLE001 -- reported below, by every rank that saw one
                err = str(e)
            builtin_failed = agree(err)
            builtin_failed_kind = "load" if builtin_failed else None
            if not builtin_failed:
                box = [uid if rank == 0 else b""]
                dist.broadcast_object_list(box, src=0)
                try:
                    ctx.rccl_init(box[0], rank, world)
                except Exception as e:              # noqa: BLE001
                    err = str(e)
                builtin_failed = agree(err)
                builtin_failed_kind = "init" if builtin_failed else None
            if not builtin_failed:
                try:
                    ctx.slot_boundary("FT4", 0)     # no FT4 channel exists: nothing is finalised, the rendezvous runs with 0 frames
                    assert ctx.stats()["rendezvous_frames"] == 0 and ctx.stats()["rccl_world"] == world
                except Exception as e:              # noqa: BLE001
                    err = str(e) or repr(e)
                builtin_failed = agree(err)
                builtin_failed_kind = "gather" if builtin_failed else None
            if builtin_failed:
                if rank == 0:
                    print(f"bench.py: the built-in RCCL rendezvous did not come up [{builtin_failed_kind}] -- {builtin_failed}\n"
                          f"bench.py: this run uses the torch.distributed callback rendezvous instead; the record says so (multi_gpu.rendezvous / "
                          f"builtin_failed / builtin_failed_kind).  Re-run with NCCL_DEBUG=WARN for RCCL's own account.", file=sys.stderr)
                rendezvous = "torch"
        if rendezvous != "builtin":
            shard.install_rendezvous(ctx, dev)  # cwslg_slot_boundary[_end] ends in torch.distributed's all-reduce (callback form)
    SYNC_ARGS = (1.5, 200, 200, 3000)                  # jt9 -8 defaults used by the reference: syncmin 1.5, 200..3000 Hz (-H highestdecodefreq)
    if args.sync:
        ctx.enable_sync(True, *SYNC_ARGS)
    ring_blocks = SLOT_SAMPLES // IQ_LEN + 2 + (SLOT_SAMPLES % IQ_LEN != 0)
    cap = ring_blocks * IQ_LEN
    chans, rxs, freqs = [], [], []
    t_setup = time.time()
    my_slots = list(shard.slots_of_rank(S * world, rank, world))     # contiguous block partition, no data-path collective
    assert len(my_slots) == S and my_slots[0] == rank * S
    C = max(1, args.channels_per_rx)
    assert S % C == 0, "--slots must be a multiple of --channels-per-rx"
    for r in range(S // C):
        group = my_slots[r * C:(r + 1) * C]
        fl = [slot_freq(gs) for gs in group]
        gs0 = group[0]
        if C == 1:
            tones = [fl[0] + 600.0 + 37.0 * (gs0 % 11), fl[0] + 1500.0, fl[0] + 2450.0 - 13.0 * (gs0 % 7)]
        else:
            tones = [f + 1500.0 for f in fl[:8]]       # the synthetic source carries up to eight tones: the first eight channels get one each
        rx = ctx.receiver_open(FS, IQ_LEN, 0, ring_blocks=ring_blocks)
        half = cap // 2
        ctx.push_synth(rx, 0xC0FFEE ^ gs0, half, IQ_LEN, tones_hz=tones, amp=2.0e4)   # fill the ring (no channel yet)
        ctx.push_synth(rx, 0xC0FFEE ^ gs0, cap - half, IQ_LEN, tones_hz=tones, amp=2.0e4)
        for f in fl:
            ch = ctx.channel_open(rx, f, "FT8")
            rxs.append(rx); chans.append(ch); freqs.append((f, tones, 0xC0FFEE ^ gs0))
    ctx.slot_boundary("FT8", 1)           # the reference's discarded first (partial) frame
    ctx.synchronize()
    t_setup = time.time() - t_setup

    host_t = []                                        # --host-timing: seconds the host spends inside each (asynchronous) call
    def step(k):
        a = time.perf_counter()
        ctx.ring_commit_all(SLOT_SAMPLES, IQ_LEN)     # the slot's IQ is already in HBM: bookkeeping only
        b = time.perf_counter()
        ctx.process()                                  # batched NCO mix + polyphase decimate, all slots
        c_ = time.perf_counter()
        if world > 1:
            # the rendezvous of the PREVIOUS boundary runs now, while this slot's demod launch (queued just above) keeps the GPU
            # busy: wait for that boundary's kernels, all-gather the frame count over RCCL; then queue this boundary's work
            ctx.slot_boundary_end()
            ctx.slot_boundary_begin("FT8", 15 * (k + 2))
        else:
            ctx.slot_boundary("FT8", 15 * (k + 2))     # batched peak-normalise + int16 (+ sync); frames swap
        host_t.append((b - a, c_ - b, time.perf_counter() - c_))
    def barrier():
        ctx.slot_boundary_end()                        # (N > 1) the last boundary's rendezvous
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    laps = [0]                                         # slots stepped so far (the ring is re-committed lap after lap)

    def timed_region(exact):
        """W warm-up steps, then EXACTLY K timed steps in the given arithmetic mode; returns (seconds, stats, kernel name)."""
        ctx.set_exact(exact)
        for k in range(args.warmup):
            step(laps[0]); laps[0] += 1
        barrier()
        ctx.reset_stats()
        ctx.set_timing(not args.no_spans)     # HIP events on the context stream around every kernel (+ the in-kernel clock of exact-mode launches)
        del host_t[:]
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(laps[0]); laps[0] += 1
        barrier()
        dt_ = time.perf_counter() - t0
        ctx.set_timing(False)
        if args.host_timing and rank == 0:
            for k, (ta, tb, tc) in enumerate(host_t):
                print("host step %d: ring_commit_all %.3f ms, process %.3f ms, slot_boundary %.3f ms" % (k, ta * 1e3, tb * 1e3, tc * 1e3), file=sys.stderr)
        st_ = ctx.stats()
        if world > 1:
            assert st_["rendezvous_calls"] == args.steps and st_["rendezvous_frames"] == S * world, st_
            if rendezvous == "builtin":
                assert st_["rccl_world"] == world == args.gpus, st_
            every = [None] * world
            dist.all_gather_object(every, dt_)
            st_["rank_ms_per_step"] = [d / args.steps * 1e3 for d in every]
            dt_ = max(every)                   # the job's time is the slowest rank's
        return dt_, st_, ctx.demod_kernel_name()

    def verify_against_oracle(exact):
        """Post-run check of a few slots against the oracle on the exact input the last step consumed."""
        if args.verify <= 0 or rank != 0:
            return {}
        from oracle import oracle as O
        worst = 0.0
        mism = 0
        picks = sorted({int(round(x)) for x in np.linspace(0, S - 1, min(args.verify, S))})
        for s in picks:
            f, tones, seed = freqs[s]
            ring = O.synth_iq(seed, cap, FS, tones_hz=tones, amp=2.0e4)    # ring content (sample index = ring index)
            start = ((laps[0] - 1) * SLOT_SAMPLES) % cap
            idx = (start + np.arange(SLOT_SAMPLES)) % cap
            iq = ring[idx]
            oc = O.Channel("FT8", FS, IQ_LEN, f)
            oc.boundary(1); oc.boundary(2)           # discard, then emit an empty frame -> fresh demodulator
            oc.push_stream(iq)
            ref = oc.boundary(3, want_f32=True)
            a, nv = ctx.fetch_audio_f32(chans[s])
            g = ctx.fetch_frame(chans[s])
            peak = float(np.abs(ref["f32"]).max())
            worst = max(worst, float(np.abs(a.astype(np.float64) - ref["f32"]).max()) / peak)
            mism += int((g["i16"] != ref["i16"]).sum())
            if exact and not np.array_equal(a.view(np.uint32), ref["f32"].view(np.uint32)):
                print(f"PARITY FAILURE: exact mode, slot {s}: float frame differs from the reference's bits", file=sys.stderr)
                sys.exit(2)
        v = {"slots_checked": len(picks), "slots": picks, "max_rel_err": worst, "int16_mismatches": mism,
             "tolerance": 0.0 if exact else 1e-5}
        if worst > (0.0 if exact else 1e-5) or (exact and mism):
            print(f"PARITY FAILURE: {v}", file=sys.stderr)
            sys.exit(2)
        return v

    n_cu = torch.cuda.get_device_properties(local_rank).multi_processor_count or 256
    spl = S * SLOT_SAMPLES                             # complex input samples per demod launch (channel-samples: one per channel and sample)
    bps = 8.0 / C + 4.0 / 16                           # IQ is fetched once per receiver, audio written per channel

    def replay_traffic(kname):
        """HBM bytes per launch of `kname` at this slot count from the committed PMC summary (latest matching entry), or (None, None)."""
        t_, src_ = None, None
        tp = os.path.join(ROOT, "profiles", "traffic_per_launch.json")
        if os.path.isfile(tp) and C == 1:
            try:
                for tj in json.load(open(tp)).get("runs", []):
                    if tj.get("slots") == S and tj.get("kernel") == kname:
                        t_, src_ = tj.get("hbm_bytes_per_launch"), "replayed: " + tj.get("source", "profiles/traffic_per_launch.json")
            except Exception:
                t_, src_ = None, None
        return t_, src_

    def sync_geometry():
        """Search range and row pitch of the FT8 sync stage as cwslg_enable_sync derives them from the configured range (sync_host.inc), and the
        kernels the boundary launches at this channel count."""
        df = 12000.0 / 3840.0
        ia = max(1, int(round(SYNC_ARGS[2] / df)))
        ib = min(int(round(SYNC_ARGS[3] / df)), 1920 - 12)
        nbins = (ib + 13 + 31) // 32 * 32
        per_channel = S >= 2 * n_cu                   # sync_host.inc: one workgroup per channel once the chip's workgroup slots are filled
        return nbins, ib - ia + 1, ("symbol_spectra_v2_kernel + ft8_sync_chan_kernel" if per_channel else
                                    "symbol_spectra_v2_kernel + ft8_sync2d_v3_kernel + ft8_candidates_kernel")

    def roofline_sync(st_):
        """The FT8 sync stage (symbol spectra + Costas search + candidate selection) against the FP32 vector peak: it is VALU-bound
        (AI ~ 1e2 flop/B on its compulsory traffic).  Flops are the algorithm's (counted above), time is HIP events
        around the stage's launches; the bytes the stage moves through the fabric are replayed from the committed PMC summary."""
        if not args.sync or not st_["sync_launches"]:
            return None
        nbins, nsearch, kernels = sync_geometry()
        per_slot, per_transform = sync_flops_per_slot(nbins, nsearch)
        ms = st_["sync_ms"] / st_["sync_launches"]
        tfl = per_slot * S / (ms * 1e-3) / 1e12
        fabric, src, spec_insts = None, None, None
        tp = os.path.join(ROOT, "profiles", "traffic_per_launch.json")
        if os.path.isfile(tp):
            try:
                for tj in json.load(open(tp)).get("sync_runs", []):
                    if tj.get("slots") == S:
                        fabric, src = tj.get("fabric_bytes_per_boundary"), "replayed: " + tj.get("source", "")
                        spec_insts = (tj.get("valu_insts_per_launch") or {}).get("symbol_spectra_v2_kernel")
            except Exception:
                pass
        # each kernel against the resource it leans on: the spectra kernel is FP32 work (VALU-bound); the Costas search re-reads its band image
        # from LDS 63 x 8 bytes per lane and bin (42 Costas terms + 21 seven-tone sums, two adjacent lags per lane) -- reported against the LDS
        # read bandwidth, which it does NOT saturate: its wall is the per-wave dependency chain
        nl = max(1, st_["sync_launches"])
        sp_ms, se_ms = st_.get("sync_spectra_ms", 0.0) / nl, st_.get("sync_search_ms", 0.0) / nl
        lds_bytes = float(S) * nsearch * 64 * 63 * 8
        lds_peak_tbs = n_cu * 256 * 2.4e9 / 1e12                                   # ds_read_b64: 256 B / clk / CU (MI355X_MICROARCH.md, LDS) at the 2.4 GHz peak clock
        per_kernel = None
        if sp_ms > 0 and se_ms > 0:
            per_kernel = {"spectra": {"avg_ms": sp_ms, "bound": "valu", "achieved_tflops": 372 * per_transform * S / (sp_ms * 1e-3) / 1e12,
                                      "frac": 372 * per_transform * S / (sp_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                                      # the bound that moves with the arithmetic spec: the kernel's VALU instructions (SQ_INSTS_VALU of the committed PMC
                                      # summary, replayed) at the 2.35 cycles a SIMD needs per one-lane FP32 instruction with two or more resident waves
                                      # (profiles/r4_pk_issue_operands.txt), at the chip's 2.4 GHz peak clock -- the kernel runs near 2.0 GHz, so this fraction is a floor
                                      "valu_insts_per_launch": spec_insts,
                                      "issue_bound_ms_at_2p4_ghz": None if not spec_insts else spec_insts / (n_cu * 4) * 2.35 / 2.4e9 * 1e3,
                                      "issue_frac_floor": None if not spec_insts else spec_insts / (n_cu * 4) * 2.35 / 2.4e9 * 1e3 / sp_ms},
                          "search": {"avg_ms": se_ms, "bound": "lds", "lds_read_bytes": lds_bytes, "achieved_tbs": lds_bytes / (se_ms * 1e-3) / 1e12,
                                     "peak_tbs": lds_peak_tbs, "frac": lds_bytes / (se_ms * 1e-3) / 1e12 / lds_peak_tbs}}
        return {"bound": "valu", "kernels": kernels, "nbins_stored": nbins, "bins_searched": nsearch, "per_kernel": per_kernel,
                "avg_ms": ms, "transforms": 372 * S, "flop_per_transform": per_transform, "flop_per_slot": per_slot,
                "achieved_tflops": tfl, "peak_tflops": VALU_PEAK_TFLOPS, "frac": tfl / VALU_PEAK_TFLOPS,
                # round 6: the stage includes the slot's finalise (symbol_spectra_v2_kernel converts the float frame itself): it reads the float frame's
                # valid part and writes the int16 frame's (4 + 2 bytes per 12 kHz sample) instead of reading a finished int16 frame; + the lists
                "finalize_fused": st_["finalize_launches"] == 0,
                "algorithmic_bytes": S * ((180000 * 6 if st_["finalize_launches"] == 0 else 240000 * 2) + 200 * 20), "fabric_bytes": fabric, "fabric_source": src}

    def record(exact, dt_, st_, kname, ver):
        """One arithmetic mode's record: whole-job rate + the roofline object of its demod kernel."""
        launches = max(1, st_["demod_launches"])
        avg_ms = st_["demod_ms"] / launches
        achieved = bps * spl / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, traffic_source = replay_traffic(kname)
        roof = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_source, "bytes_per_sample": bps, "samples_per_launch": spl,
                "avg_launch_ms": avg_ms, "launches": st_["demod_launches"],
                "finalize_avg_ms": st_["finalize_ms"] / max(1, st_["finalize_launches"]),
                # round 6: FT8 channels with the sync stage on are finalised INSIDE symbol_spectra_v2_kernel (no separate launch: 0 here, its cost is in sync_avg_ms)
                "finalize_launches": st_["finalize_launches"],
                "sync_avg_ms": st_["sync_ms"] / max(1, st_["sync_launches"]),
                "whole_path_frac": BYTES_PER_SAMPLE_PATH * spl * args.steps / dt_ / 1e9 / HBM_PEAK_GBS}
        if exact and "exact5" in kname:
            # demod_exact5_kernel<16> (round 5; a demodulator's first outputs included: one launch per step): per tile (one 16-sample block of 32 streams = the work of 32 outputs) 16 K = 1 matrix instructions -- the
            # 32 768 un-fused products fl(y * h) -- and the FP32 pipe time of 603 one-lane instructions (15 x 32 ordered additions -- issued as 240 v_pk_add_f32
            # on register pairs, two passes each: the same pipe time at less power --, mix 48, sum * phase and workspace 64 + 3, phasor 7, ...).  The f32 MFMA occupies the SIMD's FP32 lanes: its 64 cycles and the VALU's time ADD (scripts/micro/mfma_k1.hip,
            # profiles/r5_mfma_k1.txt: gapN / both_* rows), so the bound is a one-pipe sum, priced at the clock measured inside the timed launches.
            clk_mhz = float(st_.get("demod_clock_mhz", 0.0))
            tiles = spl / 16.0 / 32.0 / (n_cu * 4.0)                                   # per SIMD and launch (the streams' 32-block warm-up not counted: it is overhead)
            mfma_cyc, valu_insts = 16 * 64, 603
            vp = {"bound": "fp32 lanes: K = 1 MFMA products + un-fused one-lane sums (one pipe: the times add, profiles/r5_mfma_k1.txt)",
                  "mfma_per_tile": 16, "cycles_per_mfma": 64, "valu_insts_per_tile": valu_insts, "cycles_per_valu_inst": 2, "of_which_issued_as_packed_pairs": 480,
                  "clock_mhz": clk_mhz or None, "clock_source": f"in-kernel s_memtime / s_memrealtime of the launch's middle workgroup over {int(st_.get('demod_clock_launches', 0))} timed launches",
                  "bound_ms": None, "frac": None}
            if clk_mhz > 0 and avg_ms > 0:
                vp["bound_ms"] = tiles * (mfma_cyc + 2 * valu_insts) / (clk_mhz * 1e3)
                vp["frac"] = vp["bound_ms"] / avg_ms
                # what two waves per SIMD retire of exactly this instruction mix with every operand in registers (both_sc, profiles/r5_mfma_k1.txt)
                vp["measured_issue_floor"] = {"cycles_per_tile": 2326, "source": "profiles/r5_mfma_k1.txt (both_sc, two waves per SIMD)", "waves_per_simd": 2,
                                              "bound_ms": tiles * 2326 / (clk_mhz * 1e3)}
                vp["measured_issue_floor"]["frac"] = vp["measured_issue_floor"]["bound_ms"] / avg_ms
            roof["valu_pipe"] = vp
        elif exact:
            # What bounds the exact kernel is not HBM but its un-fused arithmetic on the FP32 pipe: per pair of adjacent outputs
            # 33 steps x (16 taps x (multiply, add) x (Re, Im) + 4 for sum * phase) packed operations, each occupying its SIMD's vector pipe for 4
            # cycles (MI355X_MICROARCH.md: one wave's VALU instruction issues every 4 cycles; SQ_ACTIVE_INST_VALU = 1 quad-cycle per instruction in
            # profiles/r3_pmc_summary_bench4096.txt).  Clock: MEASURED inside the timed launches -- delta s_memtime / delta s_memrealtime x 100 MHz
            # read by one workgroup at the start and the end of its life (cwslg_stats.demod_clock_mhz; MI355X_MICROARCH.md, DVFS item 6).
            pk_per_pair = 33 * (16 * 2 * 2 + 4)
            clk_mhz = float(st_.get("demod_clock_mhz", 0.0))
            pipe_cycles = (spl / 16.0 / 2.0 / 64.0) * pk_per_pair * 4.0 / (n_cu * 4.0)
            vp = {"bound": "fp32 vector pipe, un-fused packed arithmetic", "packed_ops_per_output_pair": pk_per_pair, "cycles_per_packed_op": 4,
                  "clock_mhz": clk_mhz or None, "clock_source": f"in-kernel s_memtime / s_memrealtime over {int(st_.get('demod_clock_launches', 0))} timed launches",
                  "bound_ms": None, "frac": None}
            if clk_mhz > 0 and avg_ms > 0:
                vp["bound_ms"] = pipe_cycles / (clk_mhz * 1e3)
                vp["frac"] = vp["bound_ms"] / avg_ms
                # What a SIMD actually retires (scripts/micro/pk_issue.hip, profiles/r4_pk_issue.txt: bare streams of independent v_pk_mul / v_pk_add):
                # one packed operation per 5.20 cycles from one resident wave, 4.46 from two, 4.34 from three, 4.25 from four -- never 4.0.
                vp["measured_issue_floor"] = {"cycles_per_packed_op_by_waves_per_simd": {"1": 5.20, "2": 4.46, "3": 4.34, "4": 4.25},
                                              "source": "profiles/r4_pk_issue.txt", "waves_per_simd": 4 if "exact4" in kname else 2}
                cyc = 4.25 if "exact4" in kname else 4.46
                vp["measured_issue_floor"]["bound_ms"] = vp["bound_ms"] * cyc / 4.0
                vp["measured_issue_floor"]["frac"] = vp["measured_issue_floor"]["bound_ms"] / avg_ms
            roof["valu_pipe"] = vp
        else:
            roof["valu_tflops"] = 80.0 * spl / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            roof["clock_mhz"] = float(st_.get("demod_clock_mhz", 0.0)) or None     # in-kernel (s_memtime / s_memrealtime), one tile workgroup in mid-launch
        return {"mode": MODE_TEXT[exact], "value": float(world) * spl * args.steps / dt_ / 1e6, "unit": "Msamples/s", "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": dt_ / args.steps * 1e3, "roofline": roof, "roofline_sync": roofline_sync(st_), "verify": ver}

    # The headline record is the product default (exact) unless --fast; on one GPU the other mode is measured right after it on the same slots.
    primary_exact = not args.fast
    dt, st, kernel_name = timed_region(primary_exact)
    verify = verify_against_oracle(primary_exact)
    second = None
    if not args.primary_only and world == 1:
        dt2, st2, kname2 = timed_region(not primary_exact)
        ver2 = verify_against_oracle(not primary_exact)
        second = (not primary_exact, dt2, st2, kname2, ver2)

    # ---- CPU baseline: the reference's own code in the reference's shape, on this box's host cores (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        cores = host_cores()
        n_eff = SLOT_SAMPLES // IQ_LEN * IQ_LEN       # the CPU drivers push whole blocks only
        if O.have_ref():
            # kind "reference": the reference's own SSBD<float>/LowPass code (oracle/_ref, compiled from its headers)
            fn, kind = O.bench_cpu_reference, "reference"
            what = "reference SSBD.hpp/LowPass.hpp compiled -O2 -ffp-contract=off (Iterate loop + per-slot SSBD construction)"
        else:
            fn, kind = O.bench_cpu, "port"
            what = "oracle/cwsl_oracle.c -O2 -ffp-contract=off (whole path incl. prepareAudio + int16)"
        t1 = fn(1, 2)                                  # single-thread rate (reported)
        tc = fn(cores, 1)                              # calibrate the aggregate rate on `cores` threads
        slots_each = max(1, min(400, int(args.cpu_seconds / max(tc, 1e-3))))
        tN = fn(cores, slots_each) if slots_each > 1 else tc
        cpu = {"value": cores * slots_each * n_eff / tN / 1e6, "unit": "Msamples/s", "cores": cores, "kind": kind,
               "sample": f"{cores} channels x {slots_each} FT8 slots (2.88 M IQ samples each) on {cores} threads, {what}; "
                         f"single thread: {2 * n_eff / t1 / 1e6:.1f} Msamples/s; host CPU: {cpu_model()}, "
                         f"{len(os.sched_getaffinity(0))} logical CPUs visible, {cores} usable under the cgroup quota"}
        if kind == "reference" and hasattr(O, "bench_cpu_finalize"):
            # the rest of the path -- prepareAudio + int16 (Instance.cpp:294-338, 238-241), which needs <windows.h> to compile as the reference's
            # own code -- as the oracle's restatement, timed on the same threads and folded into a whole-path figure
            tf = O.bench_cpu_finalize(cores, max(4, slots_each))
            per_slot_demod = tN / slots_each
            per_slot_fin = tf / max(4, slots_each)
            cpu["whole_path_value"] = cores * n_eff / (per_slot_demod + per_slot_fin) / 1e6
            cpu["finalize_share"] = per_slot_fin / (per_slot_demod + per_slot_fin)
            cpu["sample"] += f"; whole_path_value adds prepareAudio + int16 (oracle restatement of Instance.cpp:294-338,238-241; {100 * cpu['finalize_share']:.1f}% of the path)"

    if rank == 0:
        rec = record(primary_exact, dt, st, kernel_name, verify)
        if world > 1:
            assert rec["value"] > 0 and world == args.gpus
        topo = ("private IQ stream per slot " if C == 1 else f"{C} slots share each receiver's IQ (reference topology, {S // C} receivers) ")
        out = {
            "metric": METRIC,
            "value": rec["value"], "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{S} FT8 slots/GPU x 15 s (2.88 M IQ samples) at 192 kHz, " + topo +
                                   ("(north_star: 4096 concurrent FT8 slots on ONE MI355X, all inputs resident in HBM)" if S == 4096 and world == 1 else
                                    f"(the north-star workload per GPU, x{world} GPUs: weak scaling)" if S == 4096 else
                                    "(BASELINE configs[3]: 4096 slots sharded 512 per GPU)" if S == 512 else "(--slots override)"),
                       "slots_per_gpu": S, "channels_per_receiver": C, "fs_hz": FS, "iq_block": IQ_LEN, "stages": "nco-mix+polyphase-decimate, peak-normalise+int16" + (", ft8 symbol-spectra+costas-sync+candidates" if args.sync else ""),
                       "sharding": (f"slots x{world}, no data-path collective; per slot boundary one 32-byte-per-rank RCCL all-gather inside cwslg_slot_boundary_end "
                                    f"(built-in rendezvous, cwslg_rccl_init)" if rendezvous == "builtin" else
                                    f"slots x{world}, no data-path collective; per slot boundary one 8-byte {args.dist_backend} all-reduce called back from "
                                    f"cwslg_slot_boundary_end (torch.distributed)") if world > 1 else "single GPU"},
            "mode": rec["mode"],
            "realtime_ft8_slots": rec["value"] / 0.192,
            "multi_gpu": None if world == 1 else {"rendezvous": rendezvous, "builtin_failed": builtin_failed, "builtin_failed_kind": builtin_failed_kind, "rccl_world": int(st.get("rccl_world", 0)),
                                                  "rendezvous_calls": int(st["rendezvous_calls"]), "rendezvous_frames": int(st["rendezvous_frames"]),
                                                  "rank_ms_per_step_min": min(st["rank_ms_per_step"]), "rank_ms_per_step_max": max(st["rank_ms_per_step"])},
            "roofline": rec["roofline"],
            "roofline_sync": rec["roofline_sync"],
            "verify": rec["verify"],
        }
        if C > 1:
            # with shared receivers the compulsory bytes collapse and the kernel is FP32-bound -- report it as achieved TFLOP/s of
            # useful work (80 flop per channel-sample: 6 for the mix amortised + 32 real MACs + Weaver sign; the reference spends ~340)
            avg_ms = rec["roofline"]["avg_launch_ms"]
            out["shared_topology"] = {"receivers": S // C, "channels_per_receiver": C, "channel_samples_per_launch": spl,
                                      "useful_flop_per_channel_sample": 80, "achieved_tflops": 80.0 * spl / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else None,
                                      "peak_tflops": VALU_PEAK_TFLOPS, "channel_msamples_per_s": rec["value"]}
        if second is not None:
            r2 = record(*second)
            out["exact" if second[0] else "fast"] = r2
        out["cpu_baseline"] = cpu
        out["setup_s"] = t_setup
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
