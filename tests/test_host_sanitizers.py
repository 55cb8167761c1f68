"""Host code under sanitizers (CPU only — GPU sanitizers are not available on this pool): the tap builder under
ASan + UBSan, the async dispatcher's queue / slot / thread logic under TSan and ASan + UBSan against a host-memory mock
of the 13 HIP calls it makes (tests/sanitize/mock_hip).  The "device" operations are stand-ins; nothing here computes pixels."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "sanitize")
CSRC = os.path.join(ROOT, "picha_amd", "csrc")


def _build_and_run(tmp_path, name, sources, flags, leaks=True):
    exe = str(tmp_path / name)
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-ffp-contract=off", *flags, "-I" + os.path.join(SAN, "mock_hip"), "-I" + os.path.join(ROOT, "include"),
           *sources, "-o", exe, "-lpthread"]
    subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=300)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=%d:halt_on_error=1" % (1 if leaks else 0), UBSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "ok:" in out.stdout and "Sanitizer" not in out.stderr, out.stderr[-4000:]


pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")


def test_tap_builder_under_asan_ubsan(tmp_path):
    _build_and_run(tmp_path, "taps_asan", [os.path.join(SAN, "taps_harness.cpp"), os.path.join(CSRC, "taps.cpp")],
                   ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_async_dispatcher_logic_under_sanitizers(tmp_path, san):
    flags = ["-fsanitize=" + san] + (["-fno-sanitize-recover=undefined"] if "undefined" in san else [])
    # leak check off: the per-device dispatchers (threads, pinned + device staging of three slots) are process-lifetime
    # singletons by design — freeing HIP resources from static destructors races the runtime's own teardown
    _build_and_run(tmp_path, "dispatch_" + san.split(",")[0], [os.path.join(SAN, "dispatch_harness.cpp"), os.path.join(CSRC, "dispatch.cpp")], flags,
                   leaks=False)


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_lane_pool_policy_under_sanitizers(tmp_path, san):
    """picha_amd/csrc/lane_pool.h (r03): the leasing policy of the host entry points' stream / staging / arena pool
    with a dummy lane type — exclusive leases, no frees on the return path, no more lanes than concurrent callers, lazy
    retirement through take()."""
    flags = ["-fsanitize=" + san] + (["-fno-sanitize-recover=undefined"] if "undefined" in san else [])
    _build_and_run(tmp_path, "lane_pool_" + san.split(",")[0], [os.path.join(SAN, "lane_pool_harness.cpp")], flags)
