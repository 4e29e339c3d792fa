"""GPU: the slot-boundary rendezvous BEHIND the C ABI (include/cwsl_gpu.h, multi_gpu.inc).

1. Two processes, each owning a real Context on the one GPU of the test box, shard 6 FT8 slots 3 + 3; the rendezvous
   installed through cwslg_set_boundary_rendezvous is carried by gloo (on a multi-GPU node the same hook carries RCCL:
   bench.py installs torch.distributed's "nccl" all-reduce, a C++ host calls cwslg_rccl_init).  Every
   cwslg_slot_boundary call then returns the job-wide frame count in the stats, and the frames of each shard are what
   an unsharded run produces (compared through a checksum gathered over the process group).
2. The built-in RCCL form with world size 1: librccl is opened, ncclCommInitRank / ncclAllReduce run on the context
   stream (one rank is all this box can host; N ranks differ only in the communicator's size)."""
import os
import socket
import zlib

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
FS, BLK = 192000, 2048
TOTAL, N = 6, 64 * BLK


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _slot_freq(gs):
    return -90000 + (gs * 4373) % 176000


def _run_slots(ctx, slot_ids, split=False):
    chans = []
    for gs in slot_ids:
        f = _slot_freq(gs)
        rx = ctx.receiver_open(FS, BLK, 0, ring_blocks=N // BLK + 8)
        ch = ctx.channel_open(rx, f, "FT8")
        chans.append((gs, rx, ch))
    ctx.slot_boundary("FT8", 1)                       # discarded partial slot: rendezvous carries 0 frames
    first = ctx.stats()["rendezvous_frames"]
    for gs, rx, ch in chans:
        f = _slot_freq(gs)
        ctx.push_synth(rx, 0xC0FFEE ^ gs, N, BLK, tones_hz=[f + 650.0, f + 1490.0], amp=2e4)
    if split:                                         # the boundary in two halves: the next slot's work is queued in between
        ctx.slot_boundary_begin("FT8", 16)
        with pytest.raises(Exception):
            ctx.slot_boundary_begin("FT8", 17)        # one open boundary at a time
        for gs, rx, ch in chans:
            ctx.push_synth(rx, 0xBEEF ^ gs, 4 * BLK, BLK, tones_hz=[_slot_freq(gs) + 900.0], amp=1e4)
        ctx.process()
        ctx.slot_boundary_end()
        ctx.slot_boundary_end()                       # nothing pending: no-op
    else:
        ctx.slot_boundary("FT8", 16)
    crcs = {gs: zlib.crc32(ctx.fetch_frame(ch)["i16"].tobytes()) for gs, rx, ch in chans}
    return first, ctx.stats(), crcs


def _worker(rank, world, port, q, split=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cwsl_digi_amd as P
    from cwsl_digi_amd import shard
    with P.Context(0) as ctx:                         # both ranks on GPU 0: the test box has one
        shard.install_rendezvous(ctx)
        mine = list(shard.slots_of_rank(TOTAL, rank, world))
        first, st, crcs = _run_slots(ctx, mine, split)
    gathered = [None] * world
    dist.all_gather_object(gathered, crcs)
    q.put((rank, first, st["rendezvous_calls"], st["rendezvous_frames"], st["frames_emitted"], gathered))
    dist.destroy_process_group()


@pytest.mark.parametrize("split", [False, True])
def test_two_ranks_rendezvous_through_the_c_abi(split):
    """split: cwslg_slot_boundary_begin / _end with the next slot's demodulation queued between the halves (what bench.py does
    for N > 1) -- same frames, same counts."""
    world = 2
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    ps = [mpc.Process(target=_worker, args=(r, world, port, q, split)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    import cwsl_digi_amd as P
    with P.Context(0) as ctx:                         # unsharded reference run in this process
        _, st, full = _run_slots(ctx, range(TOTAL))
        assert st["rendezvous_calls"] == 0            # no hook installed: single-GPU behaviour unchanged
    for rank, first, calls, total, emitted, gathered in res:
        assert first == 0 and calls == 2
        assert emitted == TOTAL // world and total == TOTAL     # own frames vs job-wide frames of the epoch
        merged = {}
        for part in gathered:
            merged.update(part)
        assert merged == full                          # shard r's frames == the unsharded run's, slot for slot


def test_builtin_rccl_rendezvous_world_1():
    import cwsl_digi_amd as P
    uid = P.rccl_unique_id()
    assert len(uid) == 128
    with P.Context(0) as ctx:
        ctx.rccl_init(uid, 0, 1)
        first, st, crcs = _run_slots(ctx, range(3))
        assert first == 0 and st["rendezvous_calls"] == 2 and st["rendezvous_frames"] == 3 == st["frames_emitted"]
        assert st["rccl_world"] == 1 and st["rendezvous_flags_and"] == 0
        # the flag word every rank contributes ("my inputs are exhausted"): AND over the ranks after the next boundary
        ctx.set_rendezvous_flag(1)
        ctx.slot_boundary("FT8", 99)
        assert ctx.stats()["rendezvous_flags_and"] == 1
        # a one-call boundary while a split one is open is refused, and ending it clears the state
        ctx.slot_boundary_begin("FT8", 100)
        with pytest.raises(Exception):
            ctx.slot_boundary("FT8", 101)
        ctx.slot_boundary_end()
        ctx.slot_boundary("FT8", 102)
    with P.Context(0) as ctx:                          # the two-halves form: the all-reduce runs beside the next slot's demod launch
        ctx.rccl_init(P.rccl_unique_id(), 0, 1)
        first2, st2, crcs2 = _run_slots(ctx, range(3), split=True)
        assert first2 == 0 and st2["rendezvous_calls"] == 2 and st2["rendezvous_frames"] == 3 and crcs2 == crcs


def test_builtin_rccl_communicator_life_cycle_world_1():
    """What the driver's first real N = 8 run will lean on, as far as one GPU can show it: cwslg_rccl_init brings a communicator up, 100 boundaries
    each run their all-gather on it (rendezvous_frames == the frames this -- only -- rank emitted, every time), a second cwslg_rccl_init on the same
    context replaces the communicator, cwslg_destroy releases it, and a fresh context can do all of it again (five times: a leaked
    communicator or staging buffer per cycle would show as a failure to create the next one or as growing device memory)."""
    import cwsl_digi_amd as P
    import torch
    free0 = None
    for cycle in range(5):
        with P.Context(0) as ctx:
            ctx.rccl_init(P.rccl_unique_id(), 0, 1)
            if cycle == 0:
                ctx.rccl_init(P.rccl_unique_id(), 0, 1)            # re-initialisation on a live context: the old communicator is destroyed first
            rx = ctx.receiver_open(192000, 2048, 0)
            chans = [ctx.channel_open(rx, -30000 + 9000 * k, "FT8") for k in range(4)]
            blk = np.zeros(2048, np.complex64)
            ctx.slot_boundary("FT8", 1)                            # the first, partial slot: discarded, 0 frames through the rendezvous
            assert ctx.stats()["rendezvous_frames"] == 0
            for e in range(2, 102):
                ctx.push_iq(rx, blk)
                ctx.slot_boundary("FT8", e)
                st = ctx.stats()
                assert st["rendezvous_frames"] == len(chans) and st["rccl_world"] == 1, (e, st)
            st = ctx.stats()
            assert st["rendezvous_calls"] == 101 and st["frames_emitted"] == 100 * len(chans)
            ctx.slot_boundary("FT4", 7)                            # a group without channels: the rendezvous still runs (bench.py's probe)
            assert ctx.stats()["rendezvous_calls"] == 102 and ctx.stats()["rendezvous_frames"] == 0
        free, _ = torch.cuda.mem_get_info(0)
        if cycle == 1:
            free0 = free
        if cycle == 4:
            assert free >= free0 - (64 << 20), (free0, free)       # nothing accumulates from cycle to cycle


def test_bench_py_gpus_2_launches_two_ranks_on_this_gpu():
    """The driver's N > 1 entry, end to end through bench.py itself: `python bench.py --gpus 2 ...` from a plain shell launches two ranks
    (torch.distributed.run, 127.0.0.1), each owns a real Context on the one GPU of the box (--same-device; gloo carries the rendezvous
    because RCCL refuses two ranks on one device), and rank 0's line reports n_gpus = 2 with one rendezvous per step."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--same-device", "--dist-backend", "gloo", "--slots", "64",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--verify", "2"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3
    assert line["multi_gpu"]["rendezvous_calls"] == 3 and line["multi_gpu"]["rendezvous_frames"] == 128
    assert line["mode"].startswith("exact") and line["verify"]["max_rel_err"] == 0.0 and line["verify"]["int16_mismatches"] == 0
    assert line["value"] > 0 and line["config"]["slots_per_gpu"] == 64


def test_bench_py_falls_back_loudly_when_the_builtin_communicator_cannot_come_up():
    """Two ranks on ONE device with the built-in rendezvous forced: RCCL refuses the duplicate GPU, so cwslg_rccl_init fails -- the ranks must
    agree on that, say so on stderr and in the record, and finish the run on the torch.distributed callback rendezvous (the SCALE record of an
    8-GPU node must not be lost to a communicator that did not come up)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--same-device", "--dist-backend", "gloo", "--rendezvous", "builtin",
                        "--slots", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--verify", "1", "--primary-only"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    mg = line["multi_gpu"]
    assert line["n_gpus"] == 2 and mg["rendezvous"] == "torch" and mg["builtin_failed"], mg
    assert mg["builtin_failed_kind"] == "init" and "rank 0" in mg["builtin_failed"] and "rank 1" in mg["builtin_failed"], mg   # every rank's own account
    assert mg["rendezvous_calls"] == 2 and mg["rendezvous_frames"] == 128
    assert "did not come up [init]" in p.stderr and "NCCL_DEBUG=WARN" in p.stderr
