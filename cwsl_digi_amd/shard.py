"""Multi-GPU sharding of decoder slots (independent units + one barrier).

Channels never exchange data (each Instance only reads its receiver's IQ, Instance.cpp:260-276), so
slots shard across ranks with NO data-path collective.  The only collective is the slot-boundary
rendezvous: a 4-byte all-reduce (RCCL on GPUs, gloo in the CPU tests) carrying the number of frames
each rank finalised for the epoch, so that every GPU publishes the same epoch together.
"""
import torch
import torch.distributed as dist


def slots_of_rank(n_slots_total, rank, world):
    """Contiguous block partition: rank r owns [r*S, (r+1)*S) with S = ceil(total/world) (last rank ragged)."""
    per = (n_slots_total + world - 1) // world
    lo = min(rank * per, n_slots_total)
    hi = min(lo + per, n_slots_total)
    return range(lo, hi)


def rank_of_slot(slot, n_slots_total, world):
    per = (n_slots_total + world - 1) // world
    return slot // per


def slot_boundary_rendezvous(frames_this_rank, device=None):
    """Sum of frames finalised across ranks for this slot boundary; doubles as the barrier.
    Works with any initialised backend ("nccl" = RCCL on ROCm, "gloo" on CPU)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(frames_this_rank)
    if device is None:                     # RCCL reduces device tensors, gloo host tensors
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([int(frames_this_rank)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def install_rendezvous(ctx, device=None):
    """Put the rendezvous behind the C ABI: cwslg_slot_boundary() itself then waits for its own stream, calls this
    all-reduce and records the job-wide frame count (stats()["rendezvous_frames"]) -- the host program, Python or
    C++, only ever calls slot_boundary."""
    ctx.set_boundary_rendezvous(lambda group, epoch_s, frames: slot_boundary_rendezvous(frames, device))
