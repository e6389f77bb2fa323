"""GPU: the plain-C callers of the C ABI (tools/*.c, built by `make tools`): the reference's ctest.c sequence,
its hellPerf.cpp flow on a synthetic matrix, and a CG solve -- each checks itself and exits non-zero on failure."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(name, *args):
    exe = os.path.join(ROOT, "tools", name + ".bin")
    assert os.path.exists(exe), f"{exe} missing: run `make tools`"
    out = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout


def test_ctest_sequence():
    out = _run("ctest_amd")
    assert "PASSED" in out and "hackOffsets 0 64 128 192" in out


@pytest.mark.parametrize("pattern,precision", [("banded", "d"), ("random", "s")])
def test_hellperf_flow(pattern, precision):
    out = _run("hellperf_amd", 200000, 16, pattern, 20, precision)
    assert "checksums identical: PASSED" in out


def test_cg_converges():
    out = _run("cg_amd", 128, 2000, 1e-10)
    assert "PASSED" in out
    last = [l for l in out.splitlines() if "relative residual" in l][-1]
    assert float(last.split("max |x - 1| =")[1]) < 1e-6
