#!/usr/bin/env python3
"""bench.py — throughput of picha's pixel hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N >= 1; for N > 1 it starts its own N ranks as a child torch.distributed.run)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one device-resident batch of synthetic images
(BASELINE.json configs[1]: 256 x 3840x2160 rgba -> 1920x1080, cubic).  One process per GPU; the
batch is sharded by image (every rank owns its own 256 images — weak scaling, no data-path
collective; RCCL is used only for the barrier and the max-over-ranks of the elapsed time).
Inputs are resident in HBM before the timed region starts; the metric is whole-job source Mpix/s.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     dominant kernel vs the HBM roof: algorithmic bytes per launch / measured launch time
  cpu_baseline the CPU oracle (a port of the reference algorithm) timed on this box's host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (n_images, src_w, src_h, pixel, dst_w, dst_h, opts)
    # picha.resize(img, {width, height}) — the library's default filter is cubic at filterScale 0.70
    "cfg2_4k_rgba_to_1080p_cubic": (256, 3840, 2160, "rgba", 1920, 1080, {}),
    "cfg2_4k_rgba_to_1080p_cubic_scale1": (256, 3840, 2160, "rgba", 1920, 1080, {"filter": "cubic"}),
    "cfg1_640x480_rgba_to_320x240": (256, 640, 480, "rgba", 320, 240, {}),
    "cfg3b_4k_grey_to_512_lanczos": (256, 3840, 2160, "grey", 512, 512, {"filter": "lanczos"}),
    # colour convert: dst size == src size, opts names the destination format
    "cfg3a_4k_rgb_to_grey": (256, 3840, 2160, "rgb", 3840, 2160, {"convert": "grey"}),
    # cfg3 as ONE call: colorConvert then resize chained on the device (picha_hip_convert_resize_device); algorithmic
    # bytes are those of a fused operation (source read + destination written), although the grey
    # intermediate is materialised slice by slice
    "cfg3_4k_rgb_to_grey_to_512_lanczos_chained": (256, 3840, 2160, "rgb", 512, 512, {"convert": "grey", "filter": "lanczos", "chain": True}),
    "cc_4k_rgba_to_rgba16": (64, 3840, 2160, "rgba", 3840, 2160, {"convert": "r16g16b16a16"}),
    # codec-edge pairs (LAB.md 6.3; reference index.js:37-72 supportedMap x the codecs' `encodes` lists, README.md:158-159,
    # src/pngcodec.cc:61-86): what toSupported() converts before a jpeg / webp encode and what decoders hand to resize
    "edge_jpeg_4k_rgba_to_rgb": (128, 3840, 2160, "rgba", 3840, 2160, {"convert": "rgb"}),              # tiff/png/webp decode -> jpeg encode
    "edge_jpeg_4k_rgba16_to_rgb": (64, 3840, 2160, "r16g16b16a16", 3840, 2160, {"convert": "rgb"}),      # deep png -> jpeg
    "edge_webp_4k_grey_to_rgb": (128, 3840, 2160, "grey", 3840, 2160, {"convert": "rgb"}),               # grey jpeg -> webp
    "edge_png_4k_rgba16_to_rgba": (64, 3840, 2160, "r16g16b16a16", 3840, 2160, {"convert": "rgba"}),     # deep png -> 8 bit
    "edge_jpeg_thumb_4k_rgb_to_512x288": (192, 3840, 2160, "rgb", 512, 288, {}),                         # jpeg -> thumbnail -> jpeg
    "edge_jpeg_4k_rgb_to_1080p": (192, 3840, 2160, "rgb", 1920, 1080, {}),                               # jpeg -> 1080p -> jpeg
    "cfg5_8k_rgba16_to_1080p_mitchel": (32, 7680, 4320, "r16g16b16a16", 1920, 1080, {"filter": "mitchel"}),
    # r03: resize -> colorConvert in one kernel (convert as a store epilogue), upscales
    "edge_deep_png_4k_rgba16_to_1080p_rgba": (32, 3840, 2160, "r16g16b16a16", 1920, 1080, {"convert": "rgba", "chain": True, "order": 1}),  # deep png -> 2x down -> 8-bit
    "up_960x540_rgba_to_1080p": (64, 960, 540, "rgba", 1920, 1080, {}),
    "up_960x540_r16_to_1080p": (64, 960, 540, "r16", 1920, 1080, {}),
    # r05: upscales widened — deep formats on the ring kernel, 4x on runs of four, the clipped columns produced inside the march
    "up_960x540_rgba16_to_1080p": (64, 960, 540, "r16g16b16a16", 1920, 1080, {}),
    "up_480x270_rgba_to_1080p": (256, 480, 270, "rgba", 1920, 1080, {}),
    # r03: lanczos at even integer scales (ring-aliased vertical tables: the reference's ring is one row shorter than the window)
    "lanczos_4k_rgba_to_1080p": (64, 3840, 2160, "rgba", 1920, 1080, {"filter": "lanczos"}),   # 2x: the ring-in-registers kernel
    "lanczos_4k_rgb_to_480x270": (64, 3840, 2160, "rgb", 480, 270, {"filter": "lanczos"}),      # 8x: the accumulator kernel's row list (two-pass before r03)
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable copy)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# The reference's own code (src/resize.cc:66-134 compiled unmodified, g++ -O3, 1 thread) as timed by the survey on the build
# container ("Reference CPU path, measured during the survey"): printed beside the port's figure, SURVEY.md 8(d).
REFERENCE_SURVEY = {
    "cfg2_4k_rgba_to_1080p_cubic": {"mpix_s": 90.8, "ms_per_image": 91.3},
    "cfg2_4k_rgba_to_1080p_cubic_scale1": {"mpix_s": 71.7, "ms_per_image": 115.7},
    "cfg3b_4k_grey_to_512_lanczos": {"mpix_s": 198.0, "ms_per_image": 41.8},
    "cfg5_8k_rgba16_to_1080p_mitchel": {"mpix_s": 75.4, "ms_per_image": 440.0},
    "cfg1_640x480_rgba_to_320x240": {"mpix_s": 114.0, "ms_per_image": 2.69},
}
REFERENCE_SURVEY_HW = "1 thread, 1 of 8 vCPU Intel Xeon 2.1 GHz (KVM guest, the survey container); BASELINE.md"


def _host_threads():
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return max(1, os.cpu_count() or 1)


def cpu_baseline(name):
    """The oracle (CPU port of reference src/resize.cc:66-134, specialised per pixel format like the reference's templates) on
    a bounded sample of the workload, at 1 thread, at 4 threads (libuv's default pool size = the reference's real async
    concurrency) and on every host core this process may run on."""
    import numpy as np

    import oracle
    n, sw, sh, pixel, dw, dh, opts = WORKLOADS[name]
    filt = opts.get("filter", "cubic")
    fscale = opts.get("filterScale", 1.0 if "filter" in opts else 0.70)
    threads = _host_threads()
    bpp = {"rgba": 4, "grey": 1, "r16g16b16a16": 8, "rgb": 3, "greya": 2, "r16": 2, "r16g16": 4, "r16g16b16": 6}[pixel]
    big = sw * sh * bpp > 100e6
    rng = np.random.default_rng(20250226)
    distinct = 4 if big else 16
    pool = [rng.integers(0, 256, size=sh * ((sw * bpp + 3) & ~3), dtype=np.uint8) for _ in range(distinct)]

    def run(nthreads, n_img):
        srcs = [pool[i % distinct] for i in range(n_img)]
        oracle.resize_batch_mt(srcs[:nthreads], sw, sh, pixel, dw, dh, filt, fscale, nthreads)  # warm
        t0 = time.perf_counter()
        oracle.resize_batch_mt(srcs, sw, sh, pixel, dw, dh, filt, fscale, nthreads)
        dt = time.perf_counter() - t0
        return {"value": round(n_img * sw * sh / dt / 1e6, 2), "cores": nthreads,
                "sample": "%d images (%d distinct), one image per thread on %d thread(s), %.2f s wall" % (n_img, distinct, nthreads, dt)}

    # about 10-30 s of CPU work in total (a 4K image is ~35-90 ms of one core): 24 images on one thread, 48 on four, two per
    # thread on all cores (at most 512: the destination images of one run are all live)
    n_all = threads * 2 if big else max(threads, min(512, threads * 2))
    one = run(1, 2 if big else 24)
    four = run(min(4, threads), 4 if big else 48)
    full = run(threads, n_all)
    out = {
        "value": full["value"],
        "unit": "Mpix/s",
        "cores": threads,
        "kind": "port",
        "sample": full["sample"] + " of the same workload",
        "cpu_model": _cpu_model(),
        "threads_1": one,
        "threads_4_libuv_default": four,
    }
    ref = REFERENCE_SURVEY.get(name)
    if ref:
        out["reference_survey_mpix_s"] = ref["mpix_s"]
        out["reference_survey_ms_per_image"] = ref["ms_per_image"]
        out["reference_survey_hw"] = REFERENCE_SURVEY_HW
    return out


def spawn_ranks(n, argv):
    """Run this script as n ranks under torch.distributed.run (rendezvous on 127.0.0.1, a free port) in a child process; returns its
    exit code.  Must be called before anything in this process initialises the GPU."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    print("bench.py: --gpus %d without a launcher: starting %d ranks: %s" % (n, n, " ".join(cmd)), file=sys.stderr)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # 50 x 2 ms: a 100 ms timed region (r04 review: 44 ms was thin beside a 5.7 % placement effect)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2_4k_rgba_to_1080p_cubic", choices=sorted(WORKLOADS) + ["custom"])
    ap.add_argument("--shape", default="", help="with --workload custom: n:pixel:src_w:src_h:dst_w:dst_h[:filter[:filterScale]]")
    ap.add_argument("--images", type=int, default=0, help="override images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle comparison of image 0 (profiling runs)")
    ap.add_argument("--numerics", choices=["fma", "exact"], default="exact",
                    help="fma: fused multiply-add flavour, <= 1 code value from the reference (the north star's "
                         "tolerance for the float-accumulated resample filters); exact: the reference's unfused "
                         "arithmetic, bit-identical output.  The other flavour is timed too and reported beside it.")
    ap.add_argument("--prewarm-ms", type=float, default=60.0,
                    help="setup: run the path for this long before the W warm-up steps so the GPU has left its idle clocks "
                         "(a cold process needs ~9 launches / 25 ms to reach steady state, profiles/r01_launch_series.txt)")
    ap.add_argument("--place", type=int, default=10,   # (r05: 4 until the history records — on a box with allocation history the fast placements can be candidates 7-10 only, profiles/r05_placement_history.txt)
                    help="setup: destination batches allocated and tried (the fastest is kept, the others freed); 1 = take the first allocation as it comes")
    ap.add_argument("--option", action="append", default=[], help="library knob key=value (e.g. row_chunks=4)")
    args = ap.parse_args()

    # `python bench.py --gpus N` by itself (no launcher): become N ranks.  The parent has touched no GPU (torch is not even imported yet);
    # it starts torch.distributed.run as a CHILD process — one rank per GPU over RCCL — hands it this command line unchanged and exits with
    # its code.  Under a launcher (RANK set) this is skipped and the process is one of the ranks.  The only parallelism the reference has is
    # one image per worker (src/resize.cc:362-364); here it is one shard of images per GPU.
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run --nproc-per-node %d"
                  % (args.gpus, world, args.gpus), file=sys.stderr)
        args.gpus = world
    ndev = max(1, torch.cuda.device_count())
    backend = os.environ.get("PICHA_BENCH_BACKEND", "nccl")  # "nccl" == RCCL on ROCm; "gloo" to rehearse N ranks on one GPU
    if backend == "nccl" and world > 1 and local_rank >= ndev:
        # a real (RCCL) run is one rank per GPU: more local ranks than visible devices would silently double up
        print("bench.py: LOCAL_RANK %d but only %d visible GPU(s); one rank per GPU is required with the nccl backend"
              % (local_rank, ndev), file=sys.stderr)
        sys.exit(2)
    dev = torch.device("cuda", local_rank % ndev)  # the modulo only matters for gloo rehearsals on a smaller box
    torch.cuda.set_device(dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    import picha_amd
    from picha_amd import _native as N
    from picha_amd.device import DeviceBatch, ResizePlan

    def set_opt(k, v):
        N.check(N.lib().picha_hip_set_option(k.encode(), int(v)))

    set_opt("numerics", 1 if args.numerics == "fma" else 0)
    for kv in args.option:
        k, v = kv.split("=")
        set_opt(k, v)
    numerics = "fma" if N.lib().picha_hip_get_option(b"numerics") == 1 else "exact"

    if args.workload == "custom":
        f = args.shape.split(":")
        o = {}
        if len(f) > 6 and f[6]:
            o["filter"] = f[6]
        if len(f) > 7:
            o["filterScale"] = float(f[7])
        WORKLOADS["custom"] = (int(f[0]), int(f[2]), int(f[3]), f[1], int(f[4]), int(f[5]), o)
    n, sw, sh, pixel, dw, dh, opts = WORKLOADS[args.workload]
    if args.images:
        n = args.images
    from picha_amd.shard import shard_range
    lo, hi = shard_range(n * world, rank, world)  # weak scaling: the job's batch is n images per GPU; this rank owns [lo, hi)
    assert hi - lo == n
    src = DeviceBatch(n, sw, sh, pixel, device=dev).fill_random(20250226 + 7919 * lo)
    convert_to = opts.get("convert")
    dst_pixel = convert_to or pixel
    chained = bool(opts.get("chain"))

    def make_plan(dst):
        if chained:
            from picha_amd.device import convert_resize_batch

            class _ChainPlan:
                def run(self):
                    convert_resize_batch(src, convert_to, dw, dh, filter=opts.get("filter"), filter_scale=opts.get("filterScale"), out=dst,
                                         order=int(opts.get("order", 0)))
            return _ChainPlan()
        if convert_to:
            from picha_amd.device import color_convert_batch

            class _ConvertPlan:
                def run(self):
                    color_convert_batch(src, convert_to, out=dst)
            return _ConvertPlan()
        return ResizePlan(src, dst, filter=opts.get("filter"), filter_scale=opts.get("filterScale"))

    # Setup, untimed: WHERE the destination batch lives.  The same launch into destination buffers allocated one after the other differs by 3-7 % (the first allocation
    # after the source is the slowest in every record: profiles/r04_placement.txt, r05_placement.txt), and a service allocates its output arena once — so, like one, the
    # bench allocates --place candidates, times the path into each and keeps the fastest (picha_amd.device.place_destination); every candidate is in the line.
    from picha_amd.device import place_destination
    dst, plan, placement = place_destination(lambda: DeviceBatch(n, dw, dh, dst_pixel, device=dev), make_plan, src, max(1, args.place))
    torch.cuda.synchronize()

    t_pre = time.perf_counter()  # setup, untimed: first call builds the tap tables; then leave the idle clock state
    plan.run()
    torch.cuda.synchronize()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        plan.run()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        plan.run()
    torch.cuda.synchronize()
    # what was launched (picha_hip_last_kernel): a fused convert+resize is ONE kernel; a chained pair reports the last one
    kernel = N.lib().picha_hip_last_kernel().decode()

    # per-launch durations from events on the launch stream (torch's current stream); the barrier-bracketed timing and
    # the max over ranks are picha_amd.shard.timed_region — the same code the two-rank gloo test exercises
    from picha_amd.shard import timed_region
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    step_no = [0]

    def one_step():
        if step_no[0] == 0:
            evs[0].record()
        plan.run()
        step_no[0] += 1
        evs[step_no[0]].record()

    local_s = []
    elapsed = timed_region(one_step, args.steps, torch.cuda.synchronize, dist=dist if world > 1 else None,
                           device=dev if backend == "nccl" else "cpu", local_out=local_s)
    launch_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
    avg_launch_ms = sum(launch_ms) / len(launch_ms)
    sorted_ms = sorted(launch_ms)
    median_launch_ms = sorted_ms[len(sorted_ms) // 2] if len(sorted_ms) % 2 else 0.5 * (sorted_ms[len(sorted_ms) // 2 - 1] + sorted_ms[len(sorted_ms) // 2])
    # who took part (r04): a multi-rank line carries the number of ranks that joined the all-reduce, every rank's own time and
    # the device it ran on, and an RCCL run in which two ranks report the same GPU is refused
    from picha_amd.shard import duplicate_devices, rank_report
    props = torch.cuda.get_device_properties(dev)
    dev_id = "%s/%s" % (getattr(props, "uuid", None) or getattr(props, "pci_bus_id", "?"), torch.cuda.get_device_name(dev))
    ranks_seen, per_rank_ms, per_rank_dev = rank_report(dist if world > 1 else None, dev if backend == "nccl" else "cpu", local_s[0], dev_id)
    if world > 1 and backend == "nccl" and duplicate_devices(per_rank_dev):
        if rank == 0:
            print("bench.py: ranks share a GPU under the nccl backend: %s" % duplicate_devices(per_rank_dev), file=sys.stderr)
        dist.destroy_process_group()
        sys.exit(3)
    if ranks_seen != world:
        print("bench.py: %d ranks joined the reduction but WORLD_SIZE is %d" % (ranks_seen, world), file=sys.stderr)
        sys.exit(3)

    # the other numerics flavour, timed the same way on rank 0's batch (reported, not the headline)
    other = None
    if rank == 0 and world == 1 and (chained or not convert_to):
        other_name = "exact" if numerics == "fma" else "fma"
        set_opt("numerics", 1 if other_name == "fma" else 0)
        for _ in range(max(1, args.warmup)):
            plan.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            plan.run()
        e1.record()
        torch.cuda.synchronize()
        oms = e0.elapsed_time(e1) / args.steps
        other = {"numerics": other_name, "kernel": N.lib().picha_hip_last_kernel().decode(), "avg_launch_ms": round(oms, 4)}
        other_img0 = dst.image_host(0)
        set_opt("numerics", 1 if numerics == "fma" else 0)
        plan.run()  # leave dst holding the headline flavour's output for the parity sample below
        torch.cuda.synchronize()

    # parity sample: the first TWO images of every rank's batch against the oracle (the checker, not the product);
    # rank 0 reports image 0 on its own and the maximum over both images of all ranks
    parity = parity2 = None
    if not args.no_parity:
        import numpy as np

        import oracle
        filt = opts.get("filter", "cubic")
        fscale = opts.get("filterScale", 1.0 if "filter" in opts else 0.70)

        def expected(img):
            if chained and opts.get("order"):   # resize first, then convert
                mid, _ = oracle.resize(img.data, sw, sh, pixel, dw, dh, filt, fscale)
                e, _ = oracle.color_convert(mid, dw, dh, pixel, convert_to)
            elif chained:
                mid, _ = oracle.color_convert(img.data, sw, sh, pixel, convert_to)
                e, _ = oracle.resize(mid, sw, sh, convert_to, dw, dh, filt, fscale)
            elif convert_to:
                e, _ = oracle.color_convert(img.data, sw, sh, pixel, convert_to)
            else:
                e, _ = oracle.resize(img.data, sw, sh, pixel, dw, dh, filt, fscale)
            return picha_amd.Image(width=dw, height=dh, pixel=dst_pixel, data=e).pixels().astype(np.int64)

        diffs = []
        for i in range(min(2, n)):
            b = expected(src.image_host(i))
            diffs.append(int(np.abs(dst.image_host(i).pixels().astype(np.int64) - b).max()))
            if i == 0 and other is not None:
                other["parity_max_code_diff_vs_oracle_image0"] = int(np.abs(other_img0.pixels().astype(np.int64) - b).max())
        parity, parity2 = diffs[0], max(diffs)
        if world > 1:
            t = torch.tensor([parity2], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            parity2 = int(t.item())

    if rank == 0:
        bpp = N.PIXEL_BYTES[pixel]
        src_px = n * sw * sh
        total_px = src_px * world * args.steps
        read_b = n * sw * sh * bpp
        write_b = n * dw * dh * N.PIXEL_BYTES[dst_pixel]
        alg_bytes = read_b + write_b
        ach = alg_bytes / (avg_launch_ms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel: rocprofv3 PMC passes cannot run inside this process, so the figure comes from
        # the committed profile of the same workload — and only if that profile was taken on the kernel that ran here
        def committed_traffic(kernel_name):
            """(bytes per launch scaled to this batch, provenance) from profiles/traffic_<workload>[.fma].json — whichever was measured on
            the kernel that ran; provenance alone (marked stale) if only other kernels' profiles exist"""
            stale = None
            for suffix in ("", ".fma"):
                rel = "profiles/traffic_%s%s.json" % (args.workload, suffix)
                tpath = os.path.join(ROOT, rel)
                if not os.path.exists(tpath):
                    continue
                try:
                    rec = json.load(open(tpath))
                except Exception:
                    continue
                if rec.get("kernel", "").split(" ")[0] == kernel_name.split(" ")[0]:
                    return (int(rec["hbm_bytes_per_launch"] / rec.get("images_per_launch", 256) * n),
                            {"file": rel, "kernel": rec.get("kernel"), "round": rec.get("round"), "ratio_to_algorithmic": rec.get("ratio_to_algorithmic")})
                stale = {"file": rel, "kernel": rec.get("kernel"), "stale": "measured on another kernel than the one that ran: traffic withheld"}
            return None, stale
        traffic, traffic_source = committed_traffic(kernel)

        def valu_roof(kernel_name, launch_ms):
            """The second roof (r05): VALU issue.  A wave64 f32 instruction issues every 2.05 cycles per SIMD when an even number of waves shares it (profiles/
            r05_ubench_valu_issue.txt); frac = dynamic VALU wave-instructions per launch x 2.05 / (1024 SIMDs x the kernel's cycles at its in-kernel clock).  Instruction count
            (rocprofv3 PMC SQ_INSTS_VALU) and in-kernel clock (s_memtime / s_memrealtime probe build) cannot be taken inside this process: they come from the committed
            profiles/valu_<workload>[.fma].json, and only if that was measured on the kernel that ran here."""
            for suffix in ("", ".fma"):
                rel = "profiles/valu_%s%s.json" % (args.workload, suffix)
                vpath = os.path.join(ROOT, rel)
                if not os.path.exists(vpath):
                    continue
                try:
                    rec = json.load(open(vpath))
                except Exception:
                    continue
                # the same kernel instantiation (template arguments up to the flavour); the launch form behind ",bw" (workgroup width, item list) does not change its instruction count
                if rec.get("kernel", "").split(" ")[0].split(",bw")[0] != kernel_name.split(" ")[0].split(",bw")[0]:
                    continue
                insts = rec["valu_insts_per_launch"] / rec.get("images_per_launch", 256) * n
                cpi, ghz, simds = rec.get("cycles_per_inst", 2.05), rec["inkernel_ghz"], rec.get("simds", 1024)
                issue_ms = insts * cpi / simds / (ghz * 1e6)
                return {"bound": "valu-issue", "insts_per_launch": int(insts), "cycles_per_inst": cpi, "inkernel_ghz": ghz, "simds": simds,
                        "issue_ms": round(issue_ms, 4), "frac": round(issue_ms / launch_ms, 4), "source": rel, "kernel": rec.get("kernel"), "round": rec.get("round")}
            return None
        valu = valu_roof(kernel, avg_launch_ms)
        out = {
            "metric": "Mpix/s batched 4K->1080p rgba cubic resize" if args.workload.startswith("cfg2") else "Mpix/s " + args.workload,
            "value": round(total_px / elapsed / 1e6, 1),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d x %dx%d %s -> %dx%d per GPU, opts %s (picha default = cubic, filterScale 0.70), source pixels counted"
                            % (args.workload, n, sw, sh, pixel, dw, dh, json.dumps(opts)),
                "images_per_gpu": n,
                "kernel": kernel,
                "numerics": numerics + (" (fused multiply-add; bar: <= 1 code value from the reference)" if numerics == "fma"
                                        else " (reference arithmetic; bar: bit-identical; other_numerics carries the fma flavour, which is inside the "
                                             "north star's <= 1 tolerance and is the line to hold against its HBM-read-roofline sentence)"),
                "parallelism": "image-sharded x%d" % world,
                "dst_mpix_per_s": round(n * dw * dh * world * args.steps / elapsed / 1e6, 1),
                "parity_max_code_diff_vs_oracle_image0": parity,
                "parity_max_code_diff_vs_oracle_first2_images_all_ranks": parity2,
                "setup_prewarm_ms": args.prewarm_ms,
                "ranks_seen": ranks_seen,
                "per_rank_ms": [round(x, 3) for x in per_rank_ms],
                "per_rank_device": per_rank_dev,
                # where the two batches sit relative to each other (rank 0): destination placement moves cfg2 by up to 5.7 % (profiles/r04_placement.txt), so lines
                # of different processes are comparable only with this beside them
                "dst_minus_src_mb": (dst.tensor.data_ptr() - src.tensor.data_ptr()) >> 20,
                "destination_placement": placement,   # setup: [offset from the source in MB, ms per launch] of every candidate tried, and which was kept
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(ach, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": kernel,
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": round(avg_launch_ms, 4),
                "median_launch_ms": round(median_launch_ms, 4),
                "min_launch_ms": round(sorted_ms[0], 4),
                "max_launch_ms": round(sorted_ms[-1], 4),
                "median_frac": round(alg_bytes / (median_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "read_only_frac": round(read_b / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                # which of the two roofs (HBM here, VALU issue in roofline_valu) the launch is closer to
                "tighter_roof": ("valu-issue" if valu and valu["frac"] > ach / HBM_PEAK_GBS else "hbm") if valu else None,
                # what actually limits the kernel (measured, not part of the roofline definition of SURVEY.md 8d)
                "note": ("cfg2: both flavours run at the time their loads + stores take without any arithmetic (fma 1.99 = 1.99 ms, bit-exact 2.19 vs 2.07-2.09 ms at its occupancy; "
                         "arithmetic alone 1.12 / 1.52 ms): profiles/r05_ab_cfg2_ablate.txt.  In actual fabric bytes (1.17 x the algorithmic ones) the fma flavour moves 6.2 TB/s, the "
                         "chip's measured copy rate.  VALU utilisation at 2.05 cycles per instruction is in roofline_valu (profiles/r05_issue_util.txt for the other workloads)"),
            },
        }
        if valu:
            out["roofline_valu"] = valu
        if other is not None:
            other["value_mpix_s"] = round(src_px / (other["avg_launch_ms"] * 1e-3) / 1e6, 1)
            other["roofline_frac"] = round(alg_bytes / (other["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            other["read_only_frac"] = round(read_b / (other["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            other["traffic"], other["traffic_source"] = committed_traffic(other["kernel"])
            ov = valu_roof(other["kernel"], other["avg_launch_ms"])
            if ov:
                other["roofline_valu"] = ov
            out["other_numerics"] = other
        if world == 1 and not args.no_cpu_baseline and not convert_to:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
