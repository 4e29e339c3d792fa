"""CPU, world_size 2, gloo: the N>1 path of bench.py -- slot sharding with no data-path collective and the
slot-boundary rendezvous (4-byte all-reduce).  On GPUs the same code runs on the "nccl" (= RCCL) backend."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cwsl_digi_amd import shard
    mine = list(shard.slots_of_rank(total, rank, world))
    assert all(shard.rank_of_slot(s, total, world) == rank for s in mine)
    got = []
    for epoch in range(3):
        frames = len(mine) if epoch else 0          # first boundary: every frame discarded
        got.append(shard.slot_boundary_rendezvous(frames))
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    q.put((rank, mine, got, gathered))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [4096, 1001, 1])
def test_sharding_and_boundary_rendezvous(total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    allslots = sorted(s for _, mine, _, _ in res for s in mine)
    assert allslots == list(range(total))                      # every slot owned exactly once
    for _, _, got, gathered in res:
        assert got == [0, total, total]                        # rendezvous carries the frame count
        assert sorted(s for part in gathered for s in part) == list(range(total))


def test_single_process_rendezvous_is_identity():
    from cwsl_digi_amd import shard
    assert shard.slot_boundary_rendezvous(7) == 7
    assert list(shard.slots_of_rank(10, 0, 1)) == list(range(10))


def _bench(*argv, env_extra=None, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, env=env, cwd=root)


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` from a plain shell: the parent starts two ranks through torch.distributed.run (before it imports
    torch), they form a process group on 127.0.0.1, and rank 0's line says n_gpus = 2 (--dry-run: gloo, no GPU context)."""
    import json
    p = _bench("--gpus", "2", "--dry-run", "--slots", "64")
    assert p.returncode == 0, p.stderr[-2000:]
    assert "torch.distributed.run" in p.stderr and "--nproc-per-node=2" in p.stderr
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["dry_run"] is True
    p1 = _bench("--dry-run")                                         # N = 1: no launcher, same process
    assert p1.returncode == 0 and "torch.distributed.run" not in p1.stderr
    assert json.loads(p1.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    p = _bench("--gpus", "2", "--dry-run", env_extra={"RANK": "0", "WORLD_SIZE": "4", "LOCAL_RANK": "0"})
    assert p.returncode == 2 and "WORLD_SIZE=4 but --gpus 2" in p.stderr
    p = _bench("--gpus", "1", "--dry-run", env_extra={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert p.returncode == 2


def test_bench_at_the_drivers_world_size_of_eight():
    """N = 8 first contact on the CPU: (1) `python bench.py --gpus 8 --dry-run` through bench.py's own launcher and (2) the command the driver
    runs at round end, word for word (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P
    bench.py --gpus 8 --steps K --warmup W`) plus --dry-run: eight gloo ranks join, rank 0 alone prints the line, 4096 slots per rank
    (weak scaling: configs[3]'s 512 per GPU is `--slots 512`)."""
    import json
    import subprocess
    import sys
    p = _bench("--gpus", "8", "--dry-run", timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "--nproc-per-node=8" in p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                           # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and line["dry_run"] is True and line["scaling"] == "weak"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1", "--slots", "512", "--dry-run"]
    q = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert q.returncode == 0, q.stderr[-2000:]
    lines = [l for l in q.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and line["steps"] == 3 and line["warmup"] == 1
    assert "512 FT8 slots/GPU x 8 ranks" in line["config"]["workload"]
