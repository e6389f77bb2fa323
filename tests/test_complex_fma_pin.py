"""CPU: the oracle's complex multiply-add is the vendor header's, bit for bit.

The reference's C/Z kernels call cuCfmaf / cuCfma (hell_spmv_base.cuh:33,50); the CUDA header is not in the reference
tree, SURVEY A.1 cites ROCm's twin, hipCfmaf / hipCfma (/opt/rocm/include/hip/amd_detail/amd_hip_complex.h).  This test
compiles THAT header's functions for the host with hipcc and compares them with the oracle's c_fma / z_fma (through
orc_?axpby-free entry points: one-entry HELL SpMV with alpha = 1, beta = 0 is exactly one fma onto zero; a two-entry row
chains two) on random operands -- so the complex expression tree every C/Z parity test leans on is pinned to the header,
not to this repo's reading of it."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_api as O

SRC = r"""
#include <hip/hip_complex.h>
extern "C" void vendor_cfmaf(float* out, const float* p, const float* q, const float* r, int n) {
    for (int i = 0; i < n; ++i) {
        hipFloatComplex v = hipCfmaf(make_hipFloatComplex(p[2*i], p[2*i+1]), make_hipFloatComplex(q[2*i], q[2*i+1]),
                                     make_hipFloatComplex(r[2*i], r[2*i+1]));
        out[2*i] = v.x; out[2*i+1] = v.y;
    }
}
extern "C" void vendor_cfma(double* out, const double* p, const double* q, const double* r, int n) {
    for (int i = 0; i < n; ++i) {
        hipDoubleComplex v = hipCfma(make_hipDoubleComplex(p[2*i], p[2*i+1]), make_hipDoubleComplex(q[2*i], q[2*i+1]),
                                     make_hipDoubleComplex(r[2*i], r[2*i+1]));
        out[2*i] = v.x; out[2*i+1] = v.y;
    }
}
"""


@pytest.fixture(scope="module")
def vendor(tmp_path_factory):
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    d = tmp_path_factory.mktemp("cfma")
    src, lib = d / "cfma.cpp", d / "libcfma.so"
    src.write_text(SRC)
    # host-only compile of the vendor header; -ffp-contract=fast is hipcc's default for device code and what lets the
    # header's a*b+c expressions become fused multiply-adds, as they do in the kernels
    subprocess.run([hipcc, "-x", "c++", "-O2", "-fPIC", "-shared", "-ffp-contract=fast", "-mfma", "-D__HIP_PLATFORM_AMD__",
                    "-I/opt/rocm/include", str(src), "-o", str(lib)], check=True, capture_output=True)
    return C.CDLL(str(lib))


@pytest.mark.parametrize("letter", ["C", "Z"])
def test_oracle_complex_fma_is_the_vendor_headers(vendor, letter):
    rng = np.random.default_rng(5)
    n = 4096
    real = np.float32 if letter == "C" else np.float64
    cplx = np.complex64 if letter == "C" else np.complex128
    p, q, r = (rng.standard_normal(2 * n).astype(real) for _ in range(3))
    want = np.zeros(2 * n, real)
    fn = vendor.vendor_cfmaf if letter == "C" else vendor.vendor_cfma
    ptr = lambda a: C.c_void_p(a.ctypes.data)
    fn(ptr(want), ptr(p), ptr(q), ptr(r), n)
    # the oracle's fma(p, q, r): a HELL matrix of n rows with TWO entries per row, a[i][0] = 1, x[col0] = r (so the first
    # step leaves r: fma(1, r, 0) = r exactly for finite r) and a[i][1] = p, x[col1] = q: second step = fma(p, q, r)
    pc, qc, rc = p.view(cplx), q.view(cplx), r.view(cplx)
    rows = np.repeat(np.arange(n, dtype=np.int32), 2)
    cols = np.stack([np.arange(n), n + np.arange(n)], axis=1).reshape(-1).astype(np.int32)
    vals = np.stack([np.ones(n, cplx), pc], axis=1).reshape(-1)
    x = np.concatenate([rc, qc])
    hell = O.oracle_converters.ell_to_hell(O.oracle_converters.coo_to_ell(n, rows, cols, vals), 32)
    got = O.hell_spmv(hell, x, None, 1.0, 0.0, phases=1)
    # alpha = 1: the epilogue multiplies by (1 + 0i), which must not change a finite value's bits either
    assert got.view(real).tobytes() == want.tobytes()
