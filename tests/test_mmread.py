"""CPU: the Matrix Market front end (include/spgpu/mmread.hpp, csrc/mmread.cpp) against the reference's own reader
(src/utils/mmread.cpp on NIST mmio, compiled unmodified into oracle/_ref): same C++ symbols, same files, same arrays.
Without oracle/_ref the expectations written down below (hand-derived from the files) still run."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_api as O
from spgpu_amd import capi

libc = C.CDLL(None)
libc.fopen.restype = C.c_void_p
libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
libc.fclose.argtypes = [C.c_void_p]

PROPS = b"_Z16loadMmPropertiesPiS_S_PbS_S_P8_IO_FILE"
COO = {"f": (b"_Z17loadMmMatrixToCooPfPiS0_iiibiP8_IO_FILE", np.float32), "d": (b"_Z17loadMmMatrixToCooPdPiS0_iiibiP8_IO_FILE", np.float64),
       "i": (b"_Z17loadMmMatrixToCooPiS_S_iiibiP8_IO_FILE", np.int32)}
COO_PATTERN = b"_Z17loadMmMatrixToCooPiS_iiibiP8_IO_FILE"
VEC_D = b"_Z25loadMmVectorToDenseVectorPdiiP8_IO_FILE"
ptr, i32 = C.c_void_p, C.c_int


class Reader:
    """The C++ overloads of one implementation, resolved by their mangled names."""

    def __init__(self, resolve, label):
        self.label = label
        self.props = resolve(PROPS, C.c_bool, [ptr, ptr, ptr, ptr, ptr, ptr, ptr])
        self.coo = {k: resolve(n, i32, [ptr, ptr, ptr, i32, i32, i32, C.c_bool, i32, ptr]) for k, (n, _) in COO.items()}
        self.pattern = resolve(COO_PATTERN, i32, [ptr, ptr, i32, i32, i32, C.c_bool, i32, ptr])
        self.vec_d = resolve(VEC_D, i32, [ptr, i32, i32, ptr])

    def read(self, path, kind):
        f = libc.fopen(str(path).encode(), b"r")
        assert f
        try:
            m, n, nnz, st, ty = (i32() for _ in range(5))
            sparse = C.c_bool()
            ok = self.props(C.byref(m), C.byref(n), C.byref(nnz), C.byref(sparse), C.byref(st), C.byref(ty), f)
            if not ok:
                return dict(ok=False)
            rows, cols = np.full(max(nnz.value, 1), -7, np.int32), np.full(max(nnz.value, 1), -7, np.int32)
            p = lambda a: C.c_void_p(a.ctypes.data)
            if kind == "p":
                vals = None
                code = self.pattern(p(rows), p(cols), m.value, n.value, nnz.value, sparse.value, st.value, f)
            else:
                vals = np.full(max(nnz.value, 1), -7, COO[kind][1])
                code = self.coo[kind](p(vals), p(rows), p(cols), m.value, n.value, nnz.value, sparse.value, st.value, f)
            return dict(ok=True, m=m.value, n=n.value, nnz=nnz.value, sparse=sparse.value, storage=st.value, type=ty.value,
                        code=code, rows=rows, cols=cols, vals=vals)
        finally:
            libc.fclose(f)


def _product():
    def resolve(name, res, args):
        fn = getattr(capi.lib, name.decode())
        fn.restype, fn.argtypes = res, args
        return fn
    return Reader(resolve, "product")


def _reference():
    O.reference_converters()   # opens oracle/_ref lazily

    def resolve(name, res, args):
        addr = O.orc.orc_ref_symbol(name)
        assert addr, name
        return C.CFUNCTYPE(res, *args)(addr)
    return Reader(resolve, "reference")


FILES = {
    "real_general.mtx": "%%MatrixMarket matrix coordinate real general\n% a comment\n%another\n4 5 6\n1 1 1.5\n2 3 -2.25e1\n4 5 3\n3 1 0.125\n1 5 1e-3\n4 4 7\n",
    "real_symmetric.mtx": "%%MatrixMarket MATRIX Coordinate Real Symmetric\n3 3 4\n1 1 2.0\n2 1 -1.0\n3 2 -1.0\n3 3 0.0\n",
    "integer.mtx": "%%MatrixMarket matrix coordinate integer general\n\n3 2 3\n1 2 7\n3 1 -4\n2 2 9\n",
    "pattern.mtx": "%%MatrixMarket matrix coordinate pattern general\n3 3 3\n1 3\n2 2\n3 1\n",
    "size_on_later_line.mtx": "%%MatrixMarket matrix coordinate real general\n%c\n\n  \n2 2 1\n2 1 4.0\n",
    "array.mtx": "%%MatrixMarket matrix array real general\n2 2\n1.0\n2.0\n3.0\n4.0\n",
    "bad_banner.mtx": "%%NotMatrixMarket matrix coordinate real general\n1 1 1\n1 1 1.0\n",
    "forbidden_combo.mtx": "%%MatrixMarket matrix coordinate real hermitian\n1 1 1\n1 1 1.0\n",
    "short_file.mtx": "%%MatrixMarket matrix coordinate real general\n3 3 4\n1 1 1.0\n2 2 2.0\n",
}
KINDS = {"real_general.mtx": "fd", "real_symmetric.mtx": "fd", "integer.mtx": "fid", "pattern.mtx": "pd", "size_on_later_line.mtx": "d",
         "array.mtx": "d", "bad_banner.mtx": "d", "forbidden_combo.mtx": "d", "short_file.mtx": "d"}


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("mtx")
    for name, text in FILES.items():
        (d / name).write_text(text)
    return d


def _equal(a, b):
    assert a.keys() == b.keys()
    for k in a:
        if isinstance(a[k], np.ndarray):
            assert a[k].tobytes() == b[k].tobytes(), k
        else:
            assert a[k] == b[k], k


@pytest.mark.skipif(not O.reference_available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name", sorted(FILES))
def test_same_as_reference_reader(files, name, capfd):
    if name == "array.mtx":
        pytest.skip("the reference's mm_read_mtx_crd_size retries fscanf forever on an array file's 2-number size line; "
                    "this reader reports the file as unreadable instead (test_expected_contents)")
    prod, ref = _product(), _reference()
    for kind in KINDS[name]:
        _equal(prod.read(files / name, kind), ref.read(files / name, kind))


def test_expected_contents(files):
    r = _product()
    g = r.read(files / "real_general.mtx", "d")
    assert (g["m"], g["n"], g["nnz"], g["sparse"], g["storage"], g["type"], g["code"]) == (4, 5, 6, True, 1, 0, 0)
    assert g["rows"].tolist() == [0, 1, 3, 2, 0, 3] and g["cols"].tolist() == [0, 2, 4, 0, 4, 3]   # 1-based -> 0-based
    assert g["vals"].tolist() == [1.5, -22.5, 3.0, 0.125, 1e-3, 7.0]
    s = r.read(files / "real_symmetric.mtx", "f")
    assert s["type"] == 1 and s["vals"].dtype == np.float32 and s["code"] == 0
    i = r.read(files / "integer.mtx", "i")
    assert (i["storage"], i["code"], i["vals"].tolist()) == (0, 0, [7, -4, 9])
    assert r.read(files / "integer.mtx", "f")["code"] == 0          # float overload takes integer files (mmread.cpp:158-160)
    assert r.read(files / "integer.mtx", "d")["code"] == 1          # double overload does not (mmread.cpp:181-182)
    p = r.read(files / "pattern.mtx", "p")
    assert (p["storage"], p["code"], p["rows"].tolist(), p["cols"].tolist()) == (3, 0, [0, 1, 2], [2, 1, 0])
    assert r.read(files / "pattern.mtx", "d")["code"] == 1
    assert r.read(files / "size_on_later_line.mtx", "d")["vals"].tolist() == [4.0]
    assert not r.read(files / "bad_banner.mtx", "d")["ok"] and not r.read(files / "forbidden_combo.mtx", "d")["ok"]
    assert not r.read(files / "array.mtx", "d")["ok"]              # dense files have no "M N nz" line
    short = r.read(files / "short_file.mtx", "d")        # fewer entries than announced: the rest stays untouched
    assert short["rows"].tolist() == [0, 1, -7, -7]


def test_unfold_symmetric(files):
    r = _product().read(files / "real_symmetric.mtx", "d")
    p = lambda a: C.c_void_p(a.ctypes.data)
    capi.lib.spgpuMmUnfoldedSizeD.restype = i32
    total = capi.lib.spgpuMmUnfoldedSizeD(p(r["vals"]), p(r["rows"]), p(r["cols"]), r["nnz"])
    assert total == 5        # (1,1) once, (2,1) and (3,2) twice, the explicit zero (3,3) dropped (mmutils.hpp:17-24)
    ur, uc, uv = np.zeros(total, np.int32), np.zeros(total, np.int32), np.zeros(total)
    capi.lib.spgpuMmUnfoldD(p(ur), p(uc), p(uv), p(r["rows"]), p(r["cols"]), p(r["vals"]), r["nnz"])
    assert ur.tolist() == [0, 1, 0, 2, 1] and uc.tolist() == [0, 0, 1, 1, 2] and uv.tolist() == [2.0, -1.0, -1.0, -1.0, -1.0]


def test_c_wrappers_and_pipeline_to_hell(files):
    """File -> COO -> ELL -> HELL through the C ABI, as hellPerf.cpp:75-152 does with a .mtx argument."""
    from spgpu_amd import formats
    out = (i32 * 6)()
    capi.lib.spgpuMmProperties.argtypes = [C.c_char_p, ptr]
    assert capi.lib.spgpuMmProperties(str(files / "real_general.mtx").encode(), out) == 1
    m, n, nnz = out[0], out[1], out[2]
    rows, cols, vals = np.zeros(nnz, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
    capi.lib.spgpuMmReadCoo.argtypes = [C.c_char_p, C.c_char, ptr, ptr, ptr]
    p = lambda a: C.c_void_p(a.ctypes.data)
    assert capi.lib.spgpuMmReadCoo(str(files / "real_general.mtx").encode(), b"d", p(vals), p(rows), p(cols)) == 0
    hell = formats.ell_to_hell(formats.coo_to_ell(m, rows, cols, vals), 32)
    x = np.arange(1.0, n + 1.0)
    z = O.hell_spmv(hell, x, None, 1.0, 0.0)
    dense = np.zeros((m, n)); dense[rows, cols] = vals
    assert np.allclose(z, dense @ x)
