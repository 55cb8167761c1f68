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


def timed_region(fn, steps, sync, dist=None, device=None):
    """Run fn() `steps` times between barriers; returns the MAX over ranks of the elapsed seconds.

    sync(): drains the local device (torch.cuda.synchronize on GPU, no-op on CPU).
    """
    import torch

    if dist is not None and dist.is_initialized():
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    if dist is not None and dist.is_initialized():
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed
