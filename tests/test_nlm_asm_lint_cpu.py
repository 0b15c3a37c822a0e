"""cpx_nlm_kernel issues its LDS reads in inline asm (`lds_read8_b32`, `lds_read5_u16`, `lds_gather5_b32`: many loads off
one address register) and waits for them in a SEPARATE asm statement (`lds_wait*`), so that the loads of a block are all in
flight before one wait.  Between the two the compiler believes the destination registers hold their values: a copy, a
spill or any other instruction it places there would read registers whose data has not landed, and SIInsertWaitcnts does
not look inside inline asm.  This test compiles csrc/cpx_track.hip to gfx950 assembly (no GPU) and checks the
property on the generated code: from an asm block that contains ds_read instructions to the next `s_waitcnt lgkmcnt(0)`,
no instruction outside asm blocks mentions one of the registers being loaded."""
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "classifier-pipeline_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_nothing_touches_a_register_between_its_asm_lds_read_and_the_wait(tmp_path):
    asm = tmp_path / "cpx_track.s"
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + os.path.join(REPO, "include"),
           "-I" + CSRC, "-mllvm", "-amdgpu-mfma-vgpr-form", "-x", "hip", "--cuda-device-only", "-S",
           os.path.join(CSRC, "cpx_track.hip"), "-o", str(asm)]
    # (the flags of csrc/Makefile for this file)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    pending, in_asm, counted, blocks, violations = set(), False, False, 0, []
    kernel = None
    for ln, raw in enumerate(asm.read_text().splitlines(), 1):
        line = raw.split(";")[0].strip() if "#ASM" not in raw else raw.strip()
        if raw.strip().startswith(".type") and "@function" in raw:
            kernel, pending = raw.split()[1].rstrip(","), set()
        if "#ASMSTART" in raw:
            in_asm, counted = True, False
            continue
        if "#ASMEND" in raw:
            in_asm = False
            continue
        if not line or line.startswith(".") or line.endswith(":"):
            continue
        op = line.split()[0]
        if in_asm:
            if op.startswith("ds_read"):
                if not counted:
                    blocks, counted = blocks + 1, True
                pending |= vregs(line.split(",")[0])       # the destination operand
            elif op == "s_waitcnt" and "lgkmcnt(0)" in line:
                pending = set()
            continue
        if op == "s_waitcnt" and "lgkmcnt(0)" in line:
            pending = set()
            continue
        if pending and (vregs(line) & pending):
            violations.append((kernel, ln, raw.strip()))
    assert blocks >= 8, "the asm LDS-read blocks of cpx_nlm_kernel were not found: %d" % blocks
    assert not violations, "instructions between an asm ds_read and its wait touch a register being loaded:\n" + "\n".join(
        "%s line %d: %s" % v for v in violations[:20])
