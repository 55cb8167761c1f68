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
    from picha_amd.shard import duplicate_devices, rank_report, shard_range, timed_region
    lo, hi = shard_range(n_images, rank, world)
    # every rank "processes" its own images (stand-in work: mark them), no exchange of pixel data
    owned = torch.zeros(n_images, dtype=torch.int64)
    owned[lo:hi] = rank + 1
    import time
    local = []
    elapsed = timed_region(lambda: time.sleep(0.01 * (rank + 1)), steps=3, sync=lambda: None, dist=dist, local_out=local)
    # the self-check fields of bench.py's multi-rank line (r04): ranks that joined, each rank's own time, each rank's device
    seen, per_rank_ms, per_rank_dev = rank_report(dist, "cpu", local[0], "fake-gpu-%d" % (rank if n_images != 7 else 0))
    # verification only (not part of the data path): ownership must partition the batch
    dist.all_reduce(owned, op=dist.ReduceOp.SUM)
    total = torch.tensor([hi - lo], dtype=torch.int64)
    dist.all_reduce(total, op=dist.ReduceOp.SUM)
    out_q.put((rank, lo, hi, elapsed, owned.tolist(), int(total.item()), seen, per_rank_ms, per_rank_dev, duplicate_devices(per_rank_dev)))
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
    (r0, lo0, hi0, e0, owned0, tot0, seen0, ms0, dev0, dup0), (r1, lo1, hi1, e1, owned1, tot1, seen1, ms1, dev1, dup1) = results
    assert seen0 == seen1 == world and ms0 == ms1 and dev0 == dev1               # every rank holds the same report
    assert ms0[1] > ms0[0] >= 3 * 10 * 0.9 and ms0[1] >= 3 * 20 * 0.9           # ... of each rank's OWN time (rank 1 sleeps twice as long)
    assert dup0 == ([] if n_images != 7 else ["fake-gpu-0"])                    # two ranks on one device are noticed
    assert lo0 == 0 and hi0 == lo1 and hi1 == n_images          # contiguous blocks that cover the batch
    assert tot0 == tot1 == n_images
    assert owned0 == owned1 and all(v in (1, 2) for v in owned0)  # every image owned by exactly one rank
    assert e0 == e1                                                # both ranks report the max over ranks
    assert e0 >= 3 * 0.02 * 0.9                                    # ... which is the slow rank's time


def _worker8(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from picha_amd.shard import duplicate_devices, rank_report, shard_range, timed_region
    res = {}
    for name, n_images in (("cfg4", 2048), ("cfg5", 1024)):   # BASELINE.json configs[3] / [4]: 256 / 128 images per GPU on 8 GPUs
        lo, hi = shard_range(n_images, rank, world)
        owned = torch.zeros(n_images, dtype=torch.int64)
        owned[lo:hi] = 1
        dist.all_reduce(owned, op=dist.ReduceOp.SUM)
        res[name] = (lo, hi, bool((owned == 1).all()))
    local = []
    elapsed = timed_region(lambda: None, steps=2, sync=lambda: None, dist=dist, local_out=local)
    seen, ms, devs = rank_report(dist, "cpu", local[0], "gpu-%d" % rank)
    out_q.put((rank, res, elapsed, seen, len(ms), devs, duplicate_devices(devs)))
    dist.destroy_process_group()


def test_eight_rank_sharding_of_the_baseline_batches_and_the_rank_report():
    """The 8-GPU launch shape of BASELINE.json configs[3] (2048 x 4K rgba) and configs[4] (1024 x 8K r16g16b16a16), rehearsed with
    eight gloo ranks on CPU: contiguous blocks of 256 / 128 images that partition the batch, and the fields that make the first
    real SCALE line self-checking — ranks_seen == 8, one time and one distinct device per rank."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res, elapsed, seen, n_ms, devs, dup in results:
        assert res["cfg4"] == (256 * rank, 256 * (rank + 1), True)
        assert res["cfg5"] == (128 * rank, 128 * (rank + 1), True)
        assert seen == world and n_ms == world and devs == ["gpu-%d" % r for r in range(world)] and dup == []
    assert len({r[2] for r in results}) == 1    # one max-over-ranks time


def test_bench_gpus_n_spawns_n_ranks_before_touching_the_gpu(tmp_path):
    """`python bench.py --gpus N` without a launcher must start N ranks by itself: the parent process builds a
    torch.distributed.run command for 127.0.0.1 with --nproc-per-node N and this command line, runs it as a CHILD, returns its exit code —
    and has not imported torch (let alone initialised a GPU) when it does so.  Checked here without a GPU by standing a recorder in for the
    child's interpreter."""
    import json
    import subprocess
    rec = tmp_path / "argv.json"
    fake = tmp_path / "fakepy"
    fake.write_text("#!/bin/sh\nprintf '%s\\n' \"$@\" > " + str(rec) + "\necho \"$RANK\" >> " + str(rec) + "\nexit 7\n")
    fake.chmod(0o755)
    code = (
        "import sys, os\n"
        "sys.argv = ['bench.py', '--gpus', '4', '--steps', '2', '--warmup', '1']\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "sys.executable = %r\n"
        "try:\n"
        "    bench.main()\n"
        "except SystemExit as e:\n"
        "    print('EXIT', e.code, 'torch' in sys.modules)\n" % (ROOT, str(fake)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert "EXIT 7 False" in out.stdout, (out.stdout, out.stderr[-2000:])   # the child's code comes back; torch never imported in the parent
    argv = rec.read_text().split("\n")
    assert argv[:3] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert argv[argv.index("--nproc-per-node") + 1] == "4" and argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    tail = argv[argv.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail[:6] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    # under a launcher (RANK set) the same command line does NOT spawn again
    out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench, inspect; "
                          "src = inspect.getsource(bench.main); assert '\"RANK\" not in os.environ' in src" % ROOT],
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
