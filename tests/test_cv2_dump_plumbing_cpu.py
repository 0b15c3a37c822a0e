"""tools/cv2_dump.py and the parity tests that read its fixture, exercised WITHOUT OpenCV: the dump tool runs with
oracle/cv2_shim.py standing in for cv2 (plus the C MOG2 restatement), and tests/test_cv2_parity_cpu.py must then pass
on that fixture.  This proves nothing about OpenCV -- it keeps the tool, the fixture keys and the tests in step, so that
the day somebody runs the tool against the real cv2 the comparison works."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dump_tool_and_parity_tests_agree_on_the_fixture_format(tmp_path):
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
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_cv2_parity_cpu.py"), "-q", "-x"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0 and "8 passed" in r.stdout, r.stdout[-1500:]
