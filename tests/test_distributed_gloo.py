"""The N>1 path on CPU: two processes over gloo exercise exactly the distributed plumbing bench.py
uses on RCCL — contiguous image shards that partition the batch, barrier-bracketed timing, and the
max-over-ranks of the elapsed time.  (No pixel data crosses ranks: SURVEY.md 8e.)"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_images, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from picha_amd.shard import shard_range, timed_region
    lo, hi = shard_range(n_images, rank, world)
    # every rank "processes" its own images (stand-in work: mark them), no exchange of pixel data
    owned = torch.zeros(n_images, dtype=torch.int64)
    owned[lo:hi] = rank + 1
    import time
    elapsed = timed_region(lambda: time.sleep(0.01 * (rank + 1)), steps=3, sync=lambda: None, dist=dist)
    # verification only (not part of the data path): ownership must partition the batch
    dist.all_reduce(owned, op=dist.ReduceOp.SUM)
    total = torch.tensor([hi - lo], dtype=torch.int64)
    dist.all_reduce(total, op=dist.ReduceOp.SUM)
    out_q.put((rank, lo, hi, elapsed, owned.tolist(), int(total.item())))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [7, 256])
def test_two_rank_sharding_and_timing(n_images):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_images, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, e0, owned0, tot0), (r1, lo1, hi1, e1, owned1, tot1) = results
    assert lo0 == 0 and hi0 == lo1 and hi1 == n_images          # contiguous blocks that cover the batch
    assert tot0 == tot1 == n_images
    assert owned0 == owned1 and all(v in (1, 2) for v in owned0)  # every image owned by exactly one rank
    assert e0 == e1                                                # both ranks report the max over ranks
    assert e0 >= 3 * 0.02 * 0.9                                    # ... which is the slow rank's time
