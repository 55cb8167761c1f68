"""Image-sharded data parallelism: which images of a batch a rank owns, and how ranks agree on time.

The path shards naturally — images are independent — so there is no data-path
collective: a rank resizes its own contiguous block of the batch on its own GPU.  torch.distributed
(backend "nccl" = RCCL on ROCm, "gloo" in CPU tests) is used only for the barrier around a timed
region and the max-over-ranks of the elapsed time.
"""
import time


def shard_range(n_images, rank, world):
    """Contiguous block [lo, hi) of a batch of n_images owned by `rank` (blocks differ by <= 1 image)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_images, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def rank_report(dist, device, local_elapsed_s, device_id):
    """What makes a multi-rank line self-checking (r04): how many ranks took part (all-reduce SUM of 1 — must equal the world
    size the launcher announced), every rank's own elapsed time and the identity of the device it ran on.  Returns
    (ranks_seen, [ms per rank], [device id per rank]) on every rank; single-process: (1, [ms], [id]).  No pixel data involved."""
    import torch
    ms = local_elapsed_s * 1e3
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return 1, [ms], [device_id]
    one = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    world = dist.get_world_size()
    times = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
    dist.all_gather(times, torch.tensor([ms], dtype=torch.float64, device=device))
    ids = [None] * world
    dist.all_gather_object(ids, str(device_id))
    return int(one.item()), [float(t.item()) for t in times], ids


def duplicate_devices(ids):
    """Device ids that more than one rank reported (a mis-launched job doubling up on a GPU)."""
    seen, dup = set(), []
    for d in ids:
        if d in seen and d not in dup:
            dup.append(d)
        seen.add(d)
    return dup


def timed_region(fn, steps, sync, dist=None, device=None, local_out=None):
    """Run fn() `steps` times between barriers; returns the MAX over ranks of the elapsed seconds.

    sync(): drains the local device (torch.cuda.synchronize on GPU, no-op on CPU).  local_out: a list that receives this rank's
    own elapsed seconds (its work alone: taken before the closing barrier).
    """
    import torch

    if dist is not None and dist.is_initialized():
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    if local_out is not None:
        local_out.append(time.perf_counter() - t0)   # this rank's own work, before it waits for the others
    if dist is not None and dist.is_initialized():
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed
