"""TEST INFRASTRUCTURE: ctypes access to oracle/liboracle.so (the CPU restatement)
and, through its lazy loader, to oracle/_ref/libspgpu_ref.so (the reference's own
host converters compiled from /root/reference; see oracle/Makefile)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "liboracle.so")
REF_PATH = os.path.join(ROOT, "oracle", "_ref", "libspgpu_ref.so")

orc = C.CDLL(ORACLE_PATH)
orc.orc_sizeOf.restype = C.c_size_t
orc.orc_fnv1a64.restype = C.c_uint64
orc.orc_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
orc.orc_ref_open.argtypes = [C.c_char_p, C.POINTER(C.c_char_p)]
orc.orc_ref_symbol.restype = C.c_void_p
orc.orc_ref_symbol.argtypes = [C.c_char_p]

NP_DTYPE = {"S": np.float32, "D": np.float64, "C": np.complex64, "Z": np.complex128}
TYPE_CODE = {"S": 1, "D": 2, "C": 3, "Z": 4}
LETTER_OF = {np.dtype(v): k for k, v in NP_DTYPE.items()}
ptr, i32 = C.c_void_p, C.c_int


class FloatComplex(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class DoubleComplex(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double)]


SCALAR = {"S": C.c_float, "D": C.c_double, "C": FloatComplex, "Z": DoubleComplex}


def scalar(letter, value):
    if letter in "SD":
        return SCALAR[letter](float(value))
    v = complex(value)
    return SCALAR[letter](v.real, v.imag)


def fnv(a):
    a = np.ascontiguousarray(a)
    return "%016x" % orc.orc_fnv1a64(C.c_void_p(a.ctypes.data), a.nbytes)


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


_CONV_SIGS = {
    "computeEllRowLenghts": (None, [ptr, C.POINTER(i32), i32, i32, ptr, i32]),
    "computeEllAllocPitch": (i32, [i32]),
    "cooToEll": (None, [ptr, ptr, i32, i32, i32, i32, i32, i32, ptr, ptr, ptr, i32, i32]),
    "computeHellAllocSize": (None, [C.POINTER(i32), i32, i32, ptr]),
    "ellToHell": (None, [ptr, ptr, ptr, i32, ptr, ptr, i32, i32, ptr, i32, i32]),
    "getHdiaHacksCount": (i32, [i32, i32]),
    "computeHdiaHackOffsetsFromCoo": (None, [C.POINTER(i32), ptr, i32, i32, i32, i32, ptr, ptr, i32]),
    "cooToHdia": (None, [ptr, ptr, ptr, i32, i32, i32, i32, ptr, ptr, ptr, i32, i32]),
    "computeDiaAllocPitch": (i32, [i32]),
    "computeDiaDiagonalsCount": (i32, [i32, i32, i32, ptr, ptr]),
    "coo2dia": (None, [ptr, ptr, i32, i32, i32, i32, i32, ptr, ptr, ptr, i32, i32]),
    "computeHdiaHackOffsets": (None, [C.POINTER(i32), ptr, i32, ptr, i32, i32, i32, i32]),
    "diaToHdia": (None, [ptr, ptr, ptr, i32, ptr, ptr, i32, i32, i32, i32]),
    "ellToOell": (None, [ptr, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, i32, i32]),
}


class ConverterSet:
    """The eight converter entry points of one implementation, as Python callables."""

    def __init__(self, resolve, label):
        self.label = label
        for name, (res, args) in _CONV_SIGS.items():
            setattr(self, name, resolve(name, res, args))

    # Same driving sequence as spgpu_amd.formats (hellPerf.cpp:136-152,254-264; diaPerf.cpp:254-293).
    def coo_to_ell(self, n_rows, rows, cols, vals, coo_base=0, ell_base=0):
        rows, cols = np.ascontiguousarray(rows, np.int32), np.ascontiguousarray(cols, np.int32)
        vals = np.ascontiguousarray(vals)
        letter = LETTER_OF[vals.dtype]
        row_len = np.zeros(max(n_rows, 1), np.int32)
        max_row = i32(0)
        self.computeEllRowLenghts(_p(row_len), C.byref(max_row), n_rows, rows.size, _p(rows), coo_base)
        pitch = self.computeEllAllocPitch(n_rows)
        values = np.zeros(max(max_row.value * pitch, 1), vals.dtype)
        indices = np.zeros(max(max_row.value * pitch, 1), np.int32)
        self.cooToEll(_p(values), _p(indices), pitch, pitch, max_row.value, ell_base, n_rows, rows.size,
                      _p(rows), _p(cols), _p(vals), coo_base, TYPE_CODE[letter])
        return dict(letter=letter, rows=n_rows, values=values[:max_row.value * pitch],
                    indices=indices[:max_row.value * pitch], pitch=pitch, max_row=max_row.value,
                    row_lengths=row_len[:n_rows], base=ell_base)

    def ell_to_hell(self, ell, hack_size=32):
        n_rows = ell["rows"]
        row_len = np.ascontiguousarray(ell["row_lengths"], np.int32)
        height = i32(0)
        self.computeHellAllocSize(C.byref(height), hack_size, n_rows, _p(row_len))
        slots = hack_size * height.value
        hacks = (n_rows + hack_size - 1) // hack_size
        values = np.zeros(max(slots, 1), ell["values"].dtype)
        indices = np.zeros(max(slots, 1), np.int32)
        hack_offsets = np.zeros(max(hacks, 1), np.int32)
        ev = ell["values"] if ell["values"].size else np.zeros(1, ell["values"].dtype)
        ei = ell["indices"] if ell["indices"].size else np.zeros(1, np.int32)
        self.ellToHell(_p(values), _p(indices), _p(hack_offsets), hack_size, _p(ev), _p(ei), ell["pitch"],
                       ell["pitch"], _p(row_len), n_rows, TYPE_CODE[ell["letter"]])
        return dict(letter=ell["letter"], rows=n_rows, values=values[:slots], indices=indices[:slots],
                    hack_offsets=hack_offsets[:hacks], hack_size=hack_size, height=height.value,
                    row_lengths=row_len, base=ell["base"])

    def coo_to_hdia(self, n_rows, n_cols, rows, cols, vals, hack_size=32, coo_base=0):
        rows, cols = np.ascontiguousarray(rows, np.int32), np.ascontiguousarray(cols, np.int32)
        vals = np.ascontiguousarray(vals)
        letter = LETTER_OF[vals.dtype]
        hacks = self.getHdiaHacksCount(hack_size, n_rows)
        hack_offsets = np.zeros(hacks + 1, np.int32)
        height = i32(0)
        self.computeHdiaHackOffsetsFromCoo(C.byref(height), _p(hack_offsets), hack_size, n_rows, n_cols,
                                           rows.size, _p(rows), _p(cols), coo_base)
        values = np.zeros(max(hack_size * height.value, 1), vals.dtype)
        offsets = np.zeros(max(height.value, 1), np.int32)
        self.cooToHdia(_p(values), _p(offsets), _p(hack_offsets), hack_size, n_rows, n_cols, rows.size,
                       _p(rows), _p(cols), _p(vals), coo_base, TYPE_CODE[letter])
        return dict(letter=letter, rows=n_rows, cols=n_cols, values=values[:hack_size * height.value],
                    offsets=offsets[:height.value], hack_offsets=hack_offsets, hack_size=hack_size,
                    height=height.value)


    def coo_to_dia(self, n_rows, n_cols, rows, cols, vals, coo_base=0):
        rows, cols = np.ascontiguousarray(rows, np.int32), np.ascontiguousarray(cols, np.int32)
        vals = np.ascontiguousarray(vals)
        letter = LETTER_OF[vals.dtype]
        diags = self.computeDiaDiagonalsCount(n_rows, n_cols, rows.size, _p(rows), _p(cols))
        pitch = self.computeDiaAllocPitch(n_rows)
        values = np.zeros(max(diags * pitch, 1), vals.dtype)
        offsets = np.zeros(max(diags, 1), np.int32)
        self.coo2dia(_p(values), _p(offsets), pitch, diags, n_rows, n_cols, rows.size, _p(rows), _p(cols), _p(vals),
                     coo_base, TYPE_CODE[letter])
        return dict(letter=letter, rows=n_rows, cols=n_cols, values=values[:diags * pitch], offsets=offsets[:diags],
                    pitch=pitch, diags=diags)

    def dia_to_hdia(self, dia, hack_size=32):
        n_rows = dia["rows"]
        hacks = self.getHdiaHacksCount(hack_size, n_rows)
        hack_offsets = np.zeros(hacks + 1, np.int32)
        height = i32(0)
        dv = dia["values"] if dia["values"].size else np.zeros(1, dia["values"].dtype)
        do = dia["offsets"] if dia["offsets"].size else np.zeros(1, np.int32)
        code = TYPE_CODE[dia["letter"]]
        self.computeHdiaHackOffsets(C.byref(height), _p(hack_offsets), hack_size, _p(dv), dia["pitch"], dia["diags"],
                                    n_rows, code)
        values = np.zeros(max(hack_size * height.value, 1), dia["values"].dtype)
        offsets = np.zeros(max(height.value, 1), np.int32)
        self.diaToHdia(_p(values), _p(offsets), _p(hack_offsets), hack_size, _p(dv), _p(do), dia["pitch"], dia["diags"],
                       n_rows, code)
        return dict(letter=dia["letter"], rows=n_rows, cols=dia["cols"], values=values[:hack_size * height.value],
                    offsets=offsets[:height.value], hack_offsets=hack_offsets, hack_size=hack_size, height=height.value)

    def ell_to_oell(self, ell):
        n_rows = ell["rows"]
        r_idx, dst_rs = np.zeros(max(n_rows, 1), np.int32), np.zeros(max(n_rows, 1), np.int32)
        values, indices = np.zeros_like(ell["values"]), np.zeros_like(ell["indices"])
        one = lambda a, dt: a if a.size else np.zeros(1, dt)
        rs = np.ascontiguousarray(ell["row_lengths"], np.int32)
        self.ellToOell(_p(r_idx), _p(one(values, ell["values"].dtype)), _p(one(indices, np.int32)), _p(dst_rs),
                       _p(one(ell["values"], ell["values"].dtype)), _p(one(ell["indices"], np.int32)), _p(one(rs, np.int32)),
                       ell["pitch"], ell["pitch"], n_rows, TYPE_CODE[ell["letter"]])
        return dict(ell, values=values, indices=indices, row_lengths=dst_rs[:n_rows]), r_idx[:n_rows]


def _resolve_oracle(name, res, args):
    fn = getattr(orc, "orc_" + name)
    fn.restype, fn.argtypes = res, args
    return fn


oracle_converters = ConverterSet(_resolve_oracle, "oracle")


def reference_available():
    return os.path.exists(REF_PATH)


_ref_set = None


def reference_converters():
    """The reference's own converters (oracle/_ref), or None when that build is absent."""
    global _ref_set
    if _ref_set is None and reference_available():
        err = C.c_char_p()
        if orc.orc_ref_open(REF_PATH.encode(), C.byref(err)) != 0:
            raise RuntimeError(f"cannot open {REF_PATH}: {err.value}")

        def resolve(name, res, args):
            addr = orc.orc_ref_symbol(name.encode())
            if not addr:
                raise RuntimeError(f"{name} missing from {REF_PATH}")
            return C.CFUNCTYPE(res, *args)(addr)

        _ref_set = ConverterSet(resolve, "reference")
    return _ref_set


# ---- SpMV / level-1 oracles ---------------------------------------------------------
_LOW = {"S": "s", "D": "d", "C": "c", "Z": "z"}
for _L, _T in SCALAR.items():
    _l = _LOW[_L]
    getattr(orc, f"orc_{_l}hellspmv").argtypes = [ptr, ptr, _T, ptr, ptr, i32, ptr, ptr, ptr, i32, ptr, _T, i32, i32]
    getattr(orc, f"orc_{_l}ellspmv").argtypes = [ptr, ptr, _T, ptr, ptr, i32, i32, ptr, ptr, i32, i32, ptr, _T, i32, i32]
    getattr(orc, f"orc_{_l}hdiaspmv").argtypes = [ptr, ptr, _T, ptr, ptr, i32, ptr, i32, i32, ptr, _T]
    getattr(orc, f"orc_{_l}axpby").argtypes = [ptr, i32, _T, ptr, _T, ptr]
    getattr(orc, f"orc_{_l}dot").argtypes = [ptr, i32, ptr, ptr]
    getattr(orc, f"orc_{_l}nrm2").argtypes = [ptr, i32, ptr]
    for _n in ("hellspmv", "ellspmv", "hdiaspmv", "axpby", "dot", "nrm2"):
        getattr(orc, f"orc_{_l}{_n}").restype = None


def hell_spmv(hell, x, y, alpha, beta, r_idx=None, phases=1):
    L = hell["letter"]
    z = np.zeros(hell["rows"], NP_DTYPE[L]) if y is None else np.array(y, NP_DTYPE[L], copy=True)
    yy = None if y is None else np.ascontiguousarray(y, NP_DTYPE[L])
    ri = None if r_idx is None else np.ascontiguousarray(r_idx, np.int32)
    getattr(orc, f"orc_{_LOW[L]}hellspmv")(_p(z), _p(yy), scalar(L, alpha), _p(hell["values"]), _p(hell["indices"]),
                                           hell["hack_size"], _p(hell["hack_offsets"]), _p(hell["row_lengths"]),
                                           _p(ri), hell["rows"], _p(np.ascontiguousarray(x, NP_DTYPE[L])),
                                           scalar(L, beta), hell["base"], phases)
    return z


def ell_spmv(ell, x, y, alpha, beta, r_idx=None, phases=1, with_row_sizes=True):
    L = ell["letter"]
    z = np.zeros(ell["rows"], NP_DTYPE[L]) if y is None else np.array(y, NP_DTYPE[L], copy=True)
    yy = None if y is None else np.ascontiguousarray(y, NP_DTYPE[L])
    ri = None if r_idx is None else np.ascontiguousarray(r_idx, np.int32)
    rs = ell["row_lengths"] if with_row_sizes else None
    getattr(orc, f"orc_{_LOW[L]}ellspmv")(_p(z), _p(yy), scalar(L, alpha), _p(ell["values"]), _p(ell["indices"]),
                                          ell["pitch"], ell["pitch"], _p(rs), _p(ri), ell["max_row"], ell["rows"],
                                          _p(np.ascontiguousarray(x, NP_DTYPE[L])), scalar(L, beta), ell["base"], phases)
    return z


def hdia_spmv(hdia, x, y, alpha, beta):
    L = hdia["letter"]
    z = np.zeros(hdia["rows"], NP_DTYPE[L]) if y is None else np.array(y, NP_DTYPE[L], copy=True)
    yy = None if y is None else np.ascontiguousarray(y, NP_DTYPE[L])
    getattr(orc, f"orc_{_LOW[L]}hdiaspmv")(_p(z), _p(yy), scalar(L, alpha), _p(hdia["values"]), _p(hdia["offsets"]),
                                           hdia["hack_size"], _p(hdia["hack_offsets"]), hdia["rows"], hdia["cols"],
                                           _p(np.ascontiguousarray(x, NP_DTYPE[L])), scalar(L, beta))
    return z


def axpby(letter, n, beta, y, alpha, x):
    z = np.zeros(n, NP_DTYPE[letter])
    getattr(orc, f"orc_{_LOW[letter]}axpby")(_p(z), n, scalar(letter, beta),
                                             _p(np.ascontiguousarray(y, NP_DTYPE[letter])) if y is not None else None,
                                             scalar(letter, alpha), _p(np.ascontiguousarray(x, NP_DTYPE[letter])))
    return z


def dot(letter, a, b):
    out = np.zeros(1, NP_DTYPE[letter])
    a, b = np.ascontiguousarray(a, NP_DTYPE[letter]), np.ascontiguousarray(b, NP_DTYPE[letter])
    getattr(orc, f"orc_{_LOW[letter]}dot")(_p(out), a.size, _p(a), _p(b))
    return out[0]


def nrm2(letter, a):
    real = {"S": np.float32, "D": np.float64, "C": np.float32, "Z": np.float64}[letter]
    out = np.zeros(1, real)
    a = np.ascontiguousarray(a, NP_DTYPE[letter])
    getattr(orc, f"orc_{_LOW[letter]}nrm2")(_p(out), a.size, _p(a))
    return out[0]


orc.orc_set_threads.argtypes = [i32]
orc.orc_threads.restype = i32
for _L in "SD":
    _f = getattr(orc, f"orc_{_LOW[_L]}hellspmm")
    _f.restype = None
    _f.argtypes = [ptr, ptr, SCALAR[_L], ptr, ptr, i32, ptr, ptr, ptr, i32, ptr, SCALAR[_L], i32, i32, i32, i32]


def hell_spmm(hell, X, Y, alpha, beta, r_idx=None, in_place=False):
    """Interleaved multivectors: X is [cols, count], Y/Z are [rows, count] (C-contiguous).
    in_place: call with Z aliasing Y, as a caller accumulating into Z does (spmm.h: with beta == 1 rows without
    entries are then left untouched)."""
    L = hell["letter"]
    X = np.ascontiguousarray(X, NP_DTYPE[L])
    count = X.shape[1]
    Z = np.zeros((hell["rows"], count), NP_DTYPE[L]) if Y is None else np.array(Y, NP_DTYPE[L], copy=True, order="C")
    YY = None if Y is None else (Z if in_place else np.ascontiguousarray(Y, NP_DTYPE[L]))
    ri = None if r_idx is None else np.ascontiguousarray(r_idx, np.int32)
    getattr(orc, f"orc_{_LOW[L]}hellspmm")(_p(Z), _p(YY), scalar(L, alpha), _p(hell["values"]), _p(hell["indices"]),
                                           hell["hack_size"], _p(hell["hack_offsets"]), _p(hell["row_lengths"]), _p(ri),
                                           hell["rows"], _p(X), scalar(L, beta), hell["base"], count, count, count)
    return Z


# ---- rest of Level-1 ---------------------------------------------------------------------
REAL_DTYPE = {"S": np.float32, "D": np.float64, "C": np.float32, "Z": np.float64}
for _L, _T in SCALAR.items():
    _l = _LOW[_L]
    for _n, _a in (("scal", [ptr, i32, _T, ptr]), ("abs", [ptr, i32, _T, ptr]), ("axy", [ptr, i32, _T, ptr, ptr]),
                   ("axypbz", [ptr, i32, _T, ptr, _T, ptr, ptr]), ("gath", [ptr, i32, ptr, i32, ptr]),
                   ("scat", [ptr, i32, ptr, ptr, i32, _T]), ("setscal", [i32, i32, i32, _T, ptr]),
                   ("asum", [ptr, i32, ptr]), ("amax", [ptr, i32, ptr])):
        _f = getattr(orc, f"orc_{_l}{_n}")
        _f.restype, _f.argtypes = None, _a
orc.orc_igath.argtypes = [ptr, i32, ptr, i32, ptr]
orc.orc_iscat.argtypes = [ptr, i32, ptr, ptr, i32, i32]
orc.orc_isetscal.argtypes = [i32, i32, i32, i32, ptr]
for _f in (orc.orc_igath, orc.orc_iscat, orc.orc_isetscal):
    _f.restype = None


def _dt(letter):
    return np.int32 if letter == "I" else NP_DTYPE[letter]


def _sc(letter, v):
    return i32(int(v)) if letter == "I" else scalar(letter, v)


def level1_map(letter, op, n, alpha, x, y=None, beta=None, z=None):
    """op in scal / abs / axy / axypbz; returns the output vector."""
    dt = NP_DTYPE[letter]
    out = np.zeros(n, dt)
    x = np.ascontiguousarray(x, dt)
    fn = getattr(orc, f"orc_{_LOW[letter]}{op}")
    if op in ("scal", "abs"):
        fn(_p(out), n, scalar(letter, alpha), _p(x))
    elif op == "axy":
        fn(_p(out), n, scalar(letter, alpha), _p(x), _p(np.ascontiguousarray(y, dt)))
    else:
        fn(_p(out), n, scalar(letter, beta), _p(np.ascontiguousarray(z, dt)), scalar(letter, alpha), _p(x),
           _p(np.ascontiguousarray(y, dt)))
    return out


def gath(letter, values_in, indices, base, y):
    out = np.array(values_in, _dt(letter), copy=True)
    idx = np.ascontiguousarray(indices, np.int32)
    getattr(orc, f"orc_{letter.lower()}gath")(_p(out), idx.size, _p(idx), base, _p(np.ascontiguousarray(y, _dt(letter))))
    return out


def scat(letter, y_in, values, indices, base, beta):
    out = np.array(y_in, _dt(letter), copy=True)
    idx = np.ascontiguousarray(indices, np.int32)
    getattr(orc, f"orc_{letter.lower()}scat")(_p(out), idx.size, _p(np.ascontiguousarray(values, _dt(letter))), _p(idx), base,
                                              _sc(letter, beta))
    return out


def setscal(letter, y_in, first, last, base, val):
    out = np.array(y_in, _dt(letter), copy=True)
    getattr(orc, f"orc_{letter.lower()}setscal")(first, last, base, _sc(letter, val), _p(out))
    return out


def asum(letter, x):
    out = np.zeros(1, REAL_DTYPE[letter])
    x = np.ascontiguousarray(x, NP_DTYPE[letter])
    getattr(orc, f"orc_{_LOW[letter]}asum")(_p(out), x.size, _p(x))
    return out[0]


def amax(letter, x):
    out = np.zeros(1, REAL_DTYPE[letter])
    x = np.ascontiguousarray(x, NP_DTYPE[letter])
    getattr(orc, f"orc_{_LOW[letter]}amax")(_p(out), x.size, _p(x))
    return out[0]


for _L, _T in SCALAR.items():
    _l = _LOW[_L]
    _f = getattr(orc, f"orc_{_l}diaspmv")
    _f.restype, _f.argtypes = None, [ptr, ptr, _T, ptr, ptr, i32, i32, i32, i32, ptr, _T]
    _f = getattr(orc, f"orc_{_l}ellcsput")
    _f.restype, _f.argtypes = None, [ptr, ptr, i32, i32, ptr, i32, ptr, ptr, ptr, i32]


def dia_spmv(dia, x, y, alpha, beta):
    L = dia["letter"]
    z = np.zeros(dia["rows"], NP_DTYPE[L]) if y is None else np.array(y, NP_DTYPE[L], copy=True)
    yy = None if y is None else np.ascontiguousarray(y, NP_DTYPE[L])
    getattr(orc, f"orc_{_LOW[L]}diaspmv")(_p(z), _p(yy), scalar(L, alpha), _p(dia["values"]), _p(dia["offsets"]), dia["pitch"],
                                          dia["rows"], dia["cols"], dia["diags"], _p(np.ascontiguousarray(x, NP_DTYPE[L])),
                                          scalar(L, beta))
    return z


def ell_csput(ell, a_i, a_j, a_val, base):
    """Returns the updated ELL value array."""
    L = ell["letter"]
    out = np.array(ell["values"], copy=True)
    ai, aj = np.ascontiguousarray(a_i, np.int32), np.ascontiguousarray(a_j, np.int32)
    av = np.ascontiguousarray(a_val, NP_DTYPE[L])
    getattr(orc, f"orc_{_LOW[L]}ellcsput")(_p(out), _p(ell["indices"]), ell["pitch"], ell["pitch"], _p(ell["row_lengths"]),
                                           ai.size, _p(ai), _p(aj), _p(av), base)
    return out


# ---- the slab kernel's tail-mode summation order (default fp64 / complex-fp32 kernel) ----------
TAIL_SHAPE = {"D": dict(group_rows=128, rows_per_lane=2, step=8, tail_lanes=16, phases=1),
              "C": dict(group_rows=128, rows_per_lane=2, step=8, tail_lanes=16, phases=1),
              "S": dict(group_rows=32, rows_per_lane=4, step=16, tail_lanes=16, phases=8)}
for _L, _T in SCALAR.items():
    _f = getattr(orc, f"orc_{_LOW[_L]}spmv_tail")
    _f.restype = None
    _f.argtypes = [ptr, ptr, _T, ptr, ptr, i32, ptr, i32, i32, ptr, i32, ptr, i32, ptr, _T, i32, i32, i32, i32, i32, i32]


for _L, _T in SCALAR.items():
    _f = getattr(orc, f"orc_{_LOW[_L]}spmv_deep")
    _f.restype = None
    _f.argtypes = [ptr, ptr, _T, ptr, ptr, i32, ptr, i32, i32, ptr, i32, ptr, i32, ptr, _T, i32, i32, i32, i32, i32, i32,
                   i32, i32, i32]
    _f = getattr(orc, f"orc_{_LOW[_L]}spmv_split")
    _f.restype = None
    _f.argtypes = [ptr, ptr, _T, ptr, ptr, i32, ptr, i32, i32, ptr, i32, ptr, i32, ptr, _T, i32, i32, i32, i32, i32, i32,
                   i32, i32, i32, i32, i32]

DEEP_CAP = 256   # SPGPU_DEEP_CAP default
DEEP_KEEP = 64   # SPGPU_DEEP_KEEP default: columns of a deep sub-group the main kernel walks itself
SHARE_CHUNK = 48  # columns per item of shareSpmvKernel (every type)
# deepItemsKernel (csrc/ellpack_spmv.hip launchDeep): phases = 64 / (32 / rows per lane); items of 64 columns
DEEP_SHAPE = {"S": dict(deep_phases=8, deep_chunk=64), "D": dict(deep_phases=4, deep_chunk=64),
              "C": dict(deep_phases=4, deep_chunk=64), "Z": dict(deep_phases=2, deep_chunk=64)}


def spmv_tail(mat, x, y, alpha, beta, r_idx=None, with_row_sizes=True, group_rows=128, rows_per_lane=2, step=8,
              tail_lanes=16, phases=1, deep_cap=0, deep_phases=1, deep_chunk=1, main_chunk=0, deep_keep=None):
    """HELL (dict has hack_offsets) or ELL SpMV in the summation order of the tail-mode slab kernel; deep_cap > 0: with
    the deep split (32-row sub-groups deeper than deep_cap finished by deepSpmvKernel)."""
    L = mat["letter"]
    z = np.zeros(mat["rows"], NP_DTYPE[L]) if y is None else np.array(y, NP_DTYPE[L], copy=True)
    yy = None if y is None else np.ascontiguousarray(y, NP_DTYPE[L])
    ri = None if r_idx is None else np.ascontiguousarray(r_idx, np.int32)
    is_hell = "hack_offsets" in mat
    rs = mat["row_lengths"] if (is_hell or with_row_sizes) else None
    getattr(orc, f"orc_{_LOW[L]}spmv_split")(
        _p(z), _p(yy), scalar(L, alpha), _p(mat["values"]), _p(mat["indices"]), mat["hack_size"] if is_hell else 0,
        _p(mat["hack_offsets"]) if is_hell else None, 0 if is_hell else mat["pitch"], 0 if is_hell else mat["pitch"],
        _p(rs), 0 if is_hell else mat["max_row"], _p(ri), mat["rows"], _p(np.ascontiguousarray(x, NP_DTYPE[L])),
        scalar(L, beta), mat["base"], group_rows, rows_per_lane, step, tail_lanes, phases, deep_cap, deep_phases, deep_chunk,
        main_chunk, deep_cap if deep_keep is None or deep_keep < 0 or deep_keep > deep_cap else deep_keep)
    return z


def ragged_split(letter, step, deep_cap, asked=-1):
    """Columns per chunk of a split sub-group in raggedSpmvKernel (csrc/ragged_spmv.hip.h raggedSplit): about 96, and large
    enough that the chunk sums of a workgroup whose sub-groups are all deepCap deep fit behind the x tile of the tightest shape
    (2 048 rows / 48 KiB).  The same value for every workgroup shape and for the gather form; complex fp64 never splits."""
    most = (49152 // {"S": 4, "D": 8, "C": 8, "Z": 16}[letter]) // 2048
    if most < 2 or deep_cap <= 0 or asked == 0:
        return 0
    want = -(-(asked if asked > 0 else 96) // step) * step
    need = -(-(-(-deep_cap // most)) // step) * step
    return max(want, need)


def slab_shape(letter, form="gather", tile_shape=0, deep_cap=0, split=-1, deep_keep=None):
    """spmv_tail parameters of the kernel the library runs for (type, x form, deep split): csrc/ellpack_spmv.hip
    launchSlabFamily / launchTiled.  None for the shapes without a tail (complex fp64 outside the deep split: 2 phases)."""
    rpl = {"S": 4, "D": 2, "C": 2, "Z": 1}[letter]
    import os
    if deep_keep is None:   # what the library reads (SPGPU_DEEP_KEEP, csrc/core.c), clamped to the cap as it does
        deep_keep = int(os.environ.get("SPGPU_DEEP_KEEP", DEEP_KEEP))
    deep_keep = deep_keep if 0 <= deep_keep < deep_cap else deep_cap
    deep = dict(deep_cap=deep_cap, deep_keep=deep_keep, **DEEP_SHAPE[letter]) if deep_cap > 0 else {}
    if form == "share":    # shareSpmvKernel (csrc/share_spmv.hip.h): (sub-group, chunk) items, 48 columns per chunk, the
        phases = 2 * rpl   # chunk sums of a sub-group added in chunk order
        return dict(group_rows=32, rows_per_lane=rpl, step=phases * (2 if rpl >= 4 else 3), tail_lanes=0, phases=phases,
                    deep_cap=SHARE_CHUNK, deep_phases=phases, deep_chunk=SHARE_CHUNK)
    if form == "ragged":   # raggedSpmvKernel: one wavefront per 32-row sub-group, 64 / (32 / rpl) phases, no tail rows
        phases = 2 * rpl
        step = phases * (2 if rpl >= 4 else 3)
        return dict(group_rows=32, rows_per_lane=rpl, step=step, tail_lanes=0, phases=phases,
                    main_chunk=ragged_split(letter, step, deep_cap, split), **deep)
    if form == "xtile":
        if tile_shape == 1 and not deep:
            return dict(group_rows=32, rows_per_lane=rpl, step=4 * rpl, tail_lanes=16, phases=2 * rpl) if rpl > 1 else None
        if letter == "Z" and not deep:
            return None
        if tile_shape == 0 and not deep:      # the default tile shape adds in the order of the type's gather kernel
            return dict(TAIL_SHAPE[letter])
        return dict(group_rows=64 * rpl, rows_per_lane=rpl, step=4, tail_lanes=16, phases=1, **deep)
    if deep:   # a lane walks whole rows: 8 columns per stage for 8-byte elements, 4 otherwise
        return dict(group_rows=64 * rpl, rows_per_lane=rpl, step=8 if letter in "DC" else 4, tail_lanes=16, phases=1, **deep)
    return dict(TAIL_SHAPE[letter]) if letter in TAIL_SHAPE else None


def default_spmv(mat, x, y, alpha, beta, r_idx=None):
    """The oracle in the summation order of the library's DEFAULT kernel for this matrix's type
    (spgpu_amd/csrc/ellpack_spmv.hip launchSlabFamily): D/C one phase + tail, S 8 phases + tail, Z 2 phases."""
    L = mat["letter"]
    if r_idx is not None:
        # a row order selects the queue-driven kernel (ragged_spmv.hip.h) with the deep split: sub-groups deeper than
        # the cap are finished by the deep kernels
        return spmv_tail(mat, x, y, alpha, beta, r_idx=r_idx, **slab_shape(L, "ragged", deep_cap=DEEP_CAP))
    if L in TAIL_SHAPE:
        return spmv_tail(mat, x, y, alpha, beta, r_idx=r_idx, **TAIL_SHAPE[L])
    phases = {"Z": 2}[L]
    fn = hell_spmv if "hack_offsets" in mat else ell_spmv
    return fn(mat, x, y, alpha, beta, r_idx=r_idx, phases=phases)
