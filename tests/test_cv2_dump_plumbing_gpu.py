"""The GPU half of tests/test_cv2_dump_plumbing_cpu.py: tests/test_cv2_parity_gpu.py run on a fixture the dump tool
made with oracle/cv2_shim.py standing in for cv2 -- keeps the tool's keys and the GPU parity tests in step (it pins the
kernels against the shim, which the always-on tests do anyway; the point is that the OpenCV run will work)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpu_parity_tests_run_on_a_shim_made_fixture(tmp_path):
    from test_cv2_dump_plumbing_cpu import REPO as _  # noqa: F401  (same harness)

    out = tmp_path / "shim_fixture.npz"
    code = (
        "import sys, runpy\n"
        "sys.path.insert(0, %r)\n"
        "import numpy as np, cv2_shim, mog2_oracle\n"
        "class _BG:\n"
        "    def __init__(self, history=1000, detectShadows=False):\n"
        "        self.m = None\n"
        "    def apply(self, f):\n"
        "        if self.m is None:\n"
        "            self.m = mog2_oracle.MOG2(f.shape[1], f.shape[0])\n"
        "        return self.m.apply(f)\n"
        "    def getBackgroundImage(self):\n"
        "        return self.m.getBackgroundImage()\n"
        "cv2_shim.createBackgroundSubtractorMOG2 = _BG\n"
        "cv2_shim.__version__ = 'shim'\n"
        "sys.modules['cv2'] = cv2_shim\n"
        "sys.argv = ['cv2_dump.py', %r]\n"
        "runpy.run_path(%r, run_name='__main__')\n" % (os.path.join(REPO, "oracle"), str(out), os.path.join(REPO, "tools", "cv2_dump.py")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and out.exists(), r.stderr[-1500:]
    env = dict(os.environ, CPX_CV2_FIXTURE=str(out))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_cv2_parity_gpu.py"), "-q", "-x",
                        "-m", "gpu"], env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-1500:]
