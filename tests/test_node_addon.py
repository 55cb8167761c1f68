"""The Node.js host side: N-API addon + JS facade (picha_amd/node).  Runs the plain-node test runner,
which mirrors the reference's mocha specs for the hot path.  Skipped where node is absent."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE_DIR = os.path.join(ROOT, "picha_amd", "node")
ADDON = os.path.join(NODE_DIR, "build", "Release", "picha.node")


def _run():
    if not shutil.which("node"):
        pytest.skip("node is not installed")
    if not os.path.exists(ADDON):
        subprocess.check_call(["make", "-C", NODE_DIR, "-s"])
    return subprocess.run(["node", os.path.join(NODE_DIR, "test", "run.js")], capture_output=True, text=True, timeout=300)


def test_node_facade_argument_checks_and_image_semantics():
    r = _run()
    assert r.returncode == 0, r.stdout + r.stderr
    assert "argument errors" in r.stdout and "FAIL" not in r.stdout


@pytest.mark.gpu
def test_node_addon_on_gpu_matches_reference_fixtures():
    r = _run()
    assert r.returncode == 0, r.stdout + r.stderr
    assert "conversion to grey should match" in r.stdout and "resize: async then sync" in r.stdout
    assert "FAIL" not in r.stdout
