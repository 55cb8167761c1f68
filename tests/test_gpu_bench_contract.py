"""bench.py's one-line JSON contract (the driver parses it): keys, types and the roofline / cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    d = _run("--images", "8", "--steps", "3", "--warmup", "1", "--prewarm-ms", "5")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["unit"] == "Mpix/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["parity_max_code_diff_vs_oracle_image0"] == 0          # bit-exact default flavour
    assert d["config"]["parity_max_code_diff_vs_oracle_first2_images_all_ranks"] == 0   # first two images per device
    assert d["config"]["ranks_seen"] == 1 and len(d["config"]["per_rank_ms"]) == 1 and len(d["config"]["per_rank_device"]) == 1
    # setup: ten destination candidates tried (four until the allocation-history records), the fastest kept, all of them in the line (r05); the second roof beside the first
    pl = d["config"]["destination_placement"]
    assert len(pl["candidates"]) == 10 and 0 <= pl["kept"] < 10 and pl["candidates"][pl["kept"]][1] == min(c[1] for c in pl["candidates"])
    assert pl["candidates"][pl["kept"]][0] == d["config"]["dst_minus_src_mb"]
    assert d["roofline_valu"]["bound"] == "valu-issue" and 0 < d["roofline_valu"]["frac"] < 1 and d["roofline"]["tighter_roof"] in ("hbm", "valu-issue")
    r = d["roofline"]
    assert r["min_launch_ms"] <= r["median_launch_ms"] <= r["max_launch_ms"] and r["median_frac"] > 0
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["algorithmic_bytes_per_launch"] == 8 * (3840 * 2160 * 4 + 1920 * 1080 * 4)
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mpix/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["other_numerics"]["numerics"] == "fma" and d["other_numerics"]["parity_max_code_diff_vs_oracle_image0"] <= 1
    # value = source pixels of all timed steps / elapsed
    assert abs(d["value"] - 8 * 3840 * 2160 * 3 / (d["ms_per_step"] * 3 * 1e-3) / 1e6) / d["value"] < 0.01


def test_bench_other_workloads_run():
    for w in ("cfg3a_4k_rgb_to_grey", "cfg3_4k_rgb_to_grey_to_512_lanczos_chained", "edge_deep_png_4k_rgba16_to_1080p_rgba", "up_960x540_r16_to_1080p", "lanczos_4k_rgba_to_1080p"):
        d = _run("--workload", w, "--images", "4", "--steps", "2", "--warmup", "1", "--prewarm-ms", "0", "--no-cpu-baseline")
        assert d["value"] > 0 and d["roofline"]["frac"] > 0 and d["config"]["parity_max_code_diff_vs_oracle_image0"] == 0
        if w.startswith("edge_deep_png"):
            assert "epilogue -> rgba" in d["config"]["kernel"], d["config"]["kernel"]   # resize -> convert ran as ONE kernel
    d = _run("--workload", "custom", "--shape", "4:rgb:640:480:320:240", "--steps", "2", "--warmup", "1", "--prewarm-ms", "0",
             "--no-cpu-baseline")
    assert "resize_sym<ch3" in d["config"]["kernel"] or "resize_stream<ch3" in d["config"]["kernel"]


def test_bench_two_ranks_over_gloo_on_one_gpu():
    """The N > 1 path of bench.py itself: two ranks launched by torch.distributed.run (before any GPU call), gloo instead
    of RCCL because both ranks share the one GPU of this box.  Image sharding, the barrier-bracketed timed region
    (picha_amd.shard.timed_region), the max over ranks and rank 0's JSON line all execute."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PICHA_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--images", "8", "--steps", "3",
                          "--warmup", "1", "--prewarm-ms", "5", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "image-sharded x2"
    assert d["config"]["images_per_gpu"] == 8
    # the self-check of a multi-rank line: both ranks joined, each with its own time; under gloo they share the one GPU
    assert d["config"]["ranks_seen"] == 2 and len(d["config"]["per_rank_ms"]) == 2 and len(set(d["config"]["per_rank_device"])) == 1
    assert d["config"]["parity_max_code_diff_vs_oracle_first2_images_all_ranks"] == 0
    # whole-job value: both ranks' images over the max-over-ranks time
    assert abs(d["value"] - 2 * 8 * 3840 * 2160 * 3 / (d["ms_per_step"] * 3 * 1e-3) / 1e6) / d["value"] < 0.01
    assert d["config"]["parity_max_code_diff_vs_oracle_image0"] == 0


def test_bench_gpus_2_launches_its_own_two_ranks():
    """`python bench.py --gpus 2` with NO launcher (the form the driver uses at N = 1): bench.py starts torch.distributed.run as a child
    before it touches the GPU and the line that comes back is a two-rank line.  gloo because this box has one GPU."""
    env = dict(os.environ, PICHA_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--images", "8", "--steps", "3", "--warmup", "1",
                          "--prewarm-ms", "5", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and len(d["config"]["per_rank_ms"]) == 2
    assert d["config"]["parallelism"] == "image-sharded x2" and d["config"]["parity_max_code_diff_vs_oracle_first2_images_all_ranks"] == 0
    assert abs(d["value"] - 2 * 8 * 3840 * 2160 * 3 / (d["ms_per_step"] * 3 * 1e-3) / 1e6) / d["value"] < 0.01
