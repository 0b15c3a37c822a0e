"""The command lines the documents give for the model converters and the parity hooks must be the tools' real
interfaces (following the documented line gave an argparse error, then a silently skipped test)."""
import os
import re
import shlex
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "INTEGRATION.md", "README.md", os.path.join("tests", "test_tf_parity_gpu.py")]


def _commands(tool):
    out = []
    for d in DOCS:
        text = open(os.path.join(REPO, d)).read()
        for m in re.finditer(r"python\s+tools/%s\s+([^`#\n|(]+)" % re.escape(tool), text):
            out.append((d, m.group(1).strip()))
    return out


def _parses(tool, argv):
    """Run the tool's argument parser only (the tools import their heavy dependencies after parse_args())."""
    code = ("import sys, runpy, argparse\n"
            "orig = argparse.ArgumentParser.parse_args\n"
            "def stop(self, *a, **k):\n"
            "    ns = orig(self, *a, **k)\n"
            "    print('PARSED', sorted(vars(ns).items()))\n"
            "    raise SystemExit(0)\n"
            "argparse.ArgumentParser.parse_args = stop\n"
            "sys.argv = [%r] + %r\n"
            "runpy.run_path(%r, run_name='__main__')\n" % (tool, argv, os.path.join(REPO, "tools", tool)))
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)


def test_documented_converter_commands_parse():
    seen = 0
    for tool in ("keras_to_npz.py", "tflite_to_npz.py", "cv2_dump.py"):
        cmds = _commands(tool)
        assert cmds, "no documented command line for %s" % tool
        for doc, args in cmds:
            argv = shlex.split(args.replace("<model.keras>", "model.keras"))
            r = _parses(tool, argv)
            assert r.returncode == 0 and "PARSED" in r.stdout, (doc, args, r.stderr[-400:])
            seen += 1
    assert seen >= 5


def test_documented_environment_variables_are_the_ones_the_tests_read():
    src = open(os.path.join(REPO, "tests", "test_tf_parity_gpu.py")).read()
    read = set(re.findall(r'os\.environ(?:\.get)?\(?\[?"(CPX_[A-Z0-9_]+)"', src))
    assert read == {"CPX_TF_MODEL", "CPX_TF_IO"}
    for d in ("DESIGN.md", "INTEGRATION.md"):
        text = open(os.path.join(REPO, d)).read()
        named = set(re.findall(r"(CPX_TF_[A-Z_]+)=", text))
        assert named and named <= read, (d, named)
        # CPX_TF_MODEL is the converter's out_base, without a suffix
        for m in re.finditer(r"CPX_TF_MODEL=(\S+)", text):
            assert not m.group(1).rstrip("`").endswith(".npz"), (d, m.group(0))
    cv = open(os.path.join(REPO, "tests", "test_cv2_parity_cpu.py")).read()
    assert "CPX_CV2_FIXTURE" in cv and "CPX_CV2_FIXTURE" in open(os.path.join(REPO, "INTEGRATION.md")).read()
