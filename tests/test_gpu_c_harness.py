"""GPU: the plain-C callers of the C ABI (tools/*.c, built by `make tools`): the reference's ctest.c sequence,
its hellPerf.cpp flow (ELL, HELL, ordered ELL) on a synthetic matrix, and a CG solve -- each checks itself and exits non-zero on failure."""
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
    assert "checksums identical: PASSED" in out and "OELL checksum equal within rounding: PASSED" in out
    # the frozen leg (spgpuHellSpmvFreeze from plain C): band columns freeze and give the unfrozen checksum; scattered ones are not frozen
    assert ("frozen HELL checksum identical: PASSED" in out) == (pattern == "banded")


@pytest.mark.parametrize("pattern,precision", [("banded", "s"), ("random", "d")])
def test_hellperf_norowsize_flow(pattern, precision):
    """The reference's hellperf_norowsize_{s,d} executables (-DNO_ROW_SIZE, src/CMakeLists.txt:186-188,
    hellPerf.cpp:200-204): the ELL run with rS == NULL walks every row to maxRowSize over the converter's zero padding and
    must print the checksum of the HELL run (which has its row sizes)."""
    out = _run("hellperf_amd", 150000, 12, pattern, 10, precision, "norowsize")
    assert "ELL (rS == NULL) dot res" in out and "checksums identical: PASSED" in out


def test_cg_timing_only_flag_keeps_the_bit_checks():
    """bench.py's call: a fixed number of iterations far from convergence; the graph runs must still repeat the eager run."""
    out = _run("cg_amd", 256, 20, 1e-30, "timing")
    assert "PASSED" in out and out.count("bit-identical to the eager run") == 2


@pytest.mark.parametrize("precision", ["d", "s"])
def test_hellperf_on_a_matrix_market_file(tmp_path, precision):
    """The reference harness's real input path: a symmetric coordinate file (lower triangle stored) is read, unfolded,
    converted and run through ELL and HELL; and a general rectangular one."""
    import numpy as np
    n = 3000
    rng = np.random.default_rng(5)
    lower = [(i, j) for i in range(n) for j in {max(0, i - 7), max(0, i - 1), i} if j <= i]
    lower = sorted(set(lower))
    sym = tmp_path / "sym.mtx"
    with open(sym, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n% test matrix\n")
        f.write(f"{n} {n} {len(lower)}\n")
        for i, j in lower:
            f.write(f"{i + 1} {j + 1} {rng.standard_normal():.17g}\n")
    out = _run("hellperf_amd", sym, 5, precision)
    unfolded = 2 * len(lower) - n
    assert f"symmetric storage unfolded: {unfolded} entries" in out
    assert f"{n} rows, {n} columns, {unfolded} nnz" in out and "checksums identical: PASSED" in out
    assert "OELL checksum equal within rounding: PASSED" in out
    rect = tmp_path / "rect.mtx"
    rows, cols, per_row = 1000, 1700, 4
    with open(rect, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{rows} {cols} {rows * per_row}\n")
        for i in range(rows):
            for j in sorted(rng.choice(cols, per_row, replace=False)):
                f.write(f"{i + 1} {j + 1} {rng.standard_normal():.17g}\n")
    out = _run("hellperf_amd", rect, 5, precision)
    assert f"{rows} rows, {cols} columns, {rows * per_row} nnz" in out and "checksums identical: PASSED" in out


@pytest.mark.parametrize("m,points,precision", [(48, 7, "d"), (300, 5, "s")])
def test_diaperf_flow(m, points, precision):
    """The reference's diaPerf.cpp flow: COO -> DIA run, COO -> HDIA run, checksums dot(z,z) identical."""
    out = _run("diaperf_amd", m, points, 10, precision)
    assert "DIA and HDIA checksums identical: PASSED" in out


def test_diaperf_on_a_matrix_market_file(tmp_path):
    import numpy as np
    n = 2000
    rng = np.random.default_rng(9)
    path = tmp_path / "band.mtx"
    entries = [(i, i + off) for i in range(n) for off in (-40, -1, 0, 1, 7) if 0 <= i + off < n]
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{n} {n} {len(entries)}\n")
        for i, j in entries:
            f.write(f"{i + 1} {j + 1} {rng.standard_normal():.17g}\n")
    out = _run("diaperf_amd", path, 5, "d")
    assert f"{n} rows, {n} columns, {len(entries)} nnz" in out and "DIA 5 diagonals" in out
    assert "DIA and HDIA checksums identical: PASSED" in out


def test_cg_converges():
    out = _run("cg_amd", 128, 2000, 1e-10)
    assert "PASSED" in out and out.count("bit-identical to the eager run") == 2     # the 8-kernel graph and the fused one
    last = [l for l in out.splitlines() if "relative residual" in l][-1]
    assert float(last.split("max |x - 1| =")[1]) < 1e-6


def test_hellperf_adopts_a_ragged_matrix_market_file(tmp_path):
    """A matrix with very unequal row lengths from a .mtx file: the harness' fourth leg (spgpuHellSpmvAdopt from plain C, then the
    same spgpuDhellspmv loop without rIdx) runs on the library's ordered copy and prints the checksum of the plain run within
    rounding; the reference's own remedy -- ellToOell + rIdx, the OELL leg -- beside it."""
    import numpy as np
    n = 6000
    rng = np.random.default_rng(11)
    lengths = np.minimum(rng.zipf(1.5, size=n) + 1, 500)
    path = tmp_path / "ragged.mtx"
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{n} {n} {int(lengths.sum())}\n")
        for i in range(n):
            cols = np.sort(rng.choice(n, int(lengths[i]), replace=False))
            for j in cols:
                f.write(f"{i + 1} {j + 1} {rng.standard_normal():.17g}\n")
    out = _run("hellperf_amd", path, 5, "d")
    assert "checksums identical: PASSED" in out and "OELL checksum equal within rounding: PASSED" in out
    assert "HELL adopted dot res" in out and "adopted HELL checksum equal within rounding: PASSED" in out
