"""Host pipeline of the reference's harness, driven through the C ABI.

``hellPerf.cpp:127-317`` does: COO -> ``computeEllRowLenghts`` / ``cooToEll`` ->
``computeHellAllocSize`` / ``ellToHell`` on the CPU, ``cudaMemcpy`` to the GPU,
then ``spgpu?hellspmv``.  The functions below run the same calls of *this*
library (host converters in ``spgpu_amd/csrc/conv_*.c``) on numpy arrays and
move the result into HBM with torch (plumbing only: torch never computes).
"""
import ctypes as C

import numpy as np

from . import capi

NP_DTYPE = {"S": np.float32, "D": np.float64, "C": np.complex64, "Z": np.complex128}
LETTER_OF = {np.dtype(v): k for k, v in NP_DTYPE.items()}


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ---- host conversions (C ABI: ell_conv.h, hell_conv.h, hdia_conv.h) ---------------

def coo_to_ell(n_rows, coo_rows, coo_cols, coo_vals, coo_base=0, ell_base=0, pitch=None):
    """COO -> ELL exactly as hellPerf.cpp:136-152 drives it (arrays zeroed first)."""
    coo_rows, coo_cols = _i32(coo_rows), _i32(coo_cols)
    coo_vals = np.ascontiguousarray(coo_vals)
    letter = LETTER_OF[coo_vals.dtype]
    nnz = int(coo_rows.size)
    row_len = np.zeros(max(n_rows, 1), dtype=np.int32)
    max_row = C.c_int(0)
    capi.computeEllRowLenghts(_p(row_len), C.byref(max_row), n_rows, nnz, _p(coo_rows), coo_base)
    if pitch is None:
        pitch = capi.computeEllAllocPitch(n_rows)
    values = np.zeros(max(max_row.value * pitch, 1), dtype=coo_vals.dtype)
    indices = np.zeros(max(max_row.value * pitch, 1), dtype=np.int32)
    capi.cooToEll(_p(values), _p(indices), pitch, pitch, max_row.value, ell_base, n_rows, nnz,
                  _p(coo_rows), _p(coo_cols), _p(coo_vals), coo_base, capi.TYPE_CODE[letter])
    return dict(letter=letter, rows=n_rows, values=values[:max_row.value * pitch],
                indices=indices[:max_row.value * pitch], pitch=pitch, max_row=max_row.value,
                row_lengths=row_len[:n_rows], base=ell_base)


def ell_to_hell(ell, hack_size=32):
    """ELL -> HELL exactly as hellPerf.cpp:254-264 drives it (but arrays zeroed, not malloc'd)."""
    n_rows = ell["rows"]
    row_len = np.ascontiguousarray(ell["row_lengths"], dtype=np.int32)
    height = C.c_int(0)
    capi.computeHellAllocSize(C.byref(height), hack_size, n_rows, _p(row_len))
    slots = hack_size * height.value
    hacks = (n_rows + hack_size - 1) // hack_size
    values = np.zeros(max(slots, 1), dtype=ell["values"].dtype)
    indices = np.zeros(max(slots, 1), dtype=np.int32)
    hack_offsets = np.zeros(max(hacks, 1), dtype=np.int32)
    ev = ell["values"] if ell["values"].size else np.zeros(1, dtype=ell["values"].dtype)
    ei = ell["indices"] if ell["indices"].size else np.zeros(1, dtype=np.int32)
    capi.ellToHell(_p(values), _p(indices), _p(hack_offsets), hack_size, _p(ev), _p(ei),
                   ell["pitch"], ell["pitch"], _p(row_len), n_rows, capi.TYPE_CODE[ell["letter"]])
    return dict(letter=ell["letter"], rows=n_rows, values=values[:slots], indices=indices[:slots],
                hack_offsets=hack_offsets[:hacks], hack_size=hack_size, height=height.value,
                row_lengths=row_len, base=ell["base"])


def coo_to_hdia(n_rows, n_cols, coo_rows, coo_cols, coo_vals, hack_size=32, coo_base=0):
    """COO -> HDIA as diaPerf.cpp:254-293 drives it, but hdiaValues zeroed first."""
    coo_rows, coo_cols = _i32(coo_rows), _i32(coo_cols)
    coo_vals = np.ascontiguousarray(coo_vals)
    letter = LETTER_OF[coo_vals.dtype]
    nnz = int(coo_rows.size)
    hacks = capi.getHdiaHacksCount(hack_size, n_rows)
    hack_offsets = np.zeros(hacks + 1, dtype=np.int32)
    height = C.c_int(0)
    capi.computeHdiaHackOffsetsFromCoo(C.byref(height), _p(hack_offsets), hack_size, n_rows, n_cols, nnz,
                                       _p(coo_rows), _p(coo_cols), coo_base)
    values = np.zeros(max(hack_size * height.value, 1), dtype=coo_vals.dtype)
    offsets = np.zeros(max(height.value, 1), dtype=np.int32)
    capi.cooToHdia(_p(values), _p(offsets), _p(hack_offsets), hack_size, n_rows, n_cols, nnz,
                   _p(coo_rows), _p(coo_cols), _p(coo_vals), coo_base, capi.TYPE_CODE[letter])
    return dict(letter=letter, rows=n_rows, cols=n_cols, values=values[:hack_size * height.value],
                offsets=offsets[:height.value], hack_offsets=hack_offsets, hack_size=hack_size,
                height=height.value)


def coo_to_dia(n_rows, n_cols, coo_rows, coo_cols, coo_vals, coo_base=0):
    """COO -> DIA as diaPerf.cpp:127-160 drives it (values zeroed first)."""
    coo_rows, coo_cols = _i32(coo_rows), _i32(coo_cols)
    coo_vals = np.ascontiguousarray(coo_vals)
    letter = LETTER_OF[coo_vals.dtype]
    nnz = int(coo_rows.size)
    diags = capi.computeDiaDiagonalsCount(n_rows, n_cols, nnz, _p(coo_rows), _p(coo_cols))
    pitch = capi.computeDiaAllocPitch(n_rows)
    values = np.zeros(max(diags * pitch, 1), dtype=coo_vals.dtype)
    offsets = np.zeros(max(diags, 1), dtype=np.int32)
    capi.coo2dia(_p(values), _p(offsets), pitch, diags, n_rows, n_cols, nnz, _p(coo_rows), _p(coo_cols), _p(coo_vals),
                 coo_base, capi.TYPE_CODE[letter])
    return dict(letter=letter, rows=n_rows, cols=n_cols, values=values[:diags * pitch], offsets=offsets[:diags],
                pitch=pitch, diags=diags)


def dia_to_hdia(dia, hack_size=32):
    """DIA -> HDIA as diaPerf.cpp:254-275 drives it, with hdiaValues zeroed first."""
    n_rows = dia["rows"]
    hacks = capi.getHdiaHacksCount(hack_size, n_rows)
    hack_offsets = np.zeros(hacks + 1, dtype=np.int32)
    height = C.c_int(0)
    dv = dia["values"] if dia["values"].size else np.zeros(1, dia["values"].dtype)
    do = dia["offsets"] if dia["offsets"].size else np.zeros(1, np.int32)
    code = capi.TYPE_CODE[dia["letter"]]
    capi.computeHdiaHackOffsets(C.byref(height), _p(hack_offsets), hack_size, _p(dv), dia["pitch"], dia["diags"], n_rows, code)
    values = np.zeros(max(hack_size * height.value, 1), dtype=dia["values"].dtype)
    offsets = np.zeros(max(height.value, 1), dtype=np.int32)
    capi.diaToHdia(_p(values), _p(offsets), _p(hack_offsets), hack_size, _p(dv), _p(do), dia["pitch"], dia["diags"], n_rows, code)
    return dict(letter=dia["letter"], rows=n_rows, cols=dia["cols"], values=values[:hack_size * height.value],
                offsets=offsets[:height.value], hack_offsets=hack_offsets, hack_size=hack_size, height=height.value)


def ell_to_oell(ell):
    """ELL -> ordered ELL (rows by descending length) as hellPerf.cpp:319-332 drives it; returns (oell, rIdx)."""
    n_rows = ell["rows"]
    r_idx = np.zeros(max(n_rows, 1), dtype=np.int32)
    dst_rs = np.zeros(max(n_rows, 1), dtype=np.int32)
    values, indices = np.zeros_like(ell["values"]), np.zeros_like(ell["indices"])
    ev = ell["values"] if ell["values"].size else np.zeros(1, ell["values"].dtype)
    ei = ell["indices"] if ell["indices"].size else np.zeros(1, np.int32)
    vv = values if values.size else np.zeros(1, ell["values"].dtype)
    ii = indices if indices.size else np.zeros(1, np.int32)
    rs = np.ascontiguousarray(ell["row_lengths"], dtype=np.int32)
    capi.ellToOell(_p(r_idx), _p(vv), _p(ii), _p(dst_rs), _p(ev), _p(ei), _p(rs), ell["pitch"], ell["pitch"], n_rows,
                   capi.TYPE_CODE[ell["letter"]])
    out = dict(ell, values=values, indices=indices, row_lengths=dst_rs[:n_rows])
    return out, r_idx[:n_rows]


def oell_order(row_lengths, window=0, long_rows=0, aligned=False):
    """The host order call (ell_conv.h oellOrder, or oellOrderAligned): returns (rIdx, sorted lengths)."""
    rs = _i32(row_lengths)
    n = int(rs.size)
    r_idx, dst = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    (capi.oellOrderAligned if aligned else capi.oellOrder)(_p(r_idx), _p(dst), _p(rs if n else np.zeros(1, np.int32)), n, window, long_rows)
    return r_idx[:n], dst[:n]


def coo_to_ordered_hell_device(handle, n_rows, coo_rows, coo_cols, coo_vals, letter, hack_size=32, window=0, long_rows=0,
                               coo_base=0, hell_base=0, order=True, r_idx_given=None, aligned=False):
    """COO arrays in HBM (torch tensors) -> HELL in HBM with its rows ordered by length, all through the C ABI:
    spgpuCooRowLengthsDevice -> spgpuOellOrderDevice -> spgpuCooPermuteRowsDevice -> spgpuCooRowLengthsDevice ->
    spgpuHellPlanDevice -> spgpuCooToHellDevice.  order=False skips the ordering (plain HELL, rIdx None).
    Returns a dict of torch tensors (cM, rP, hack_offsets, rS, rIdx) plus sizes."""
    import torch
    dev = coo_rows.device
    nnz = int(coo_rows.numel())
    torch.cuda.synchronize()   # the COO arrays were filled on torch's stream; the library works on the handle's
    work = torch.empty(capi.spgpuCooConvertWorkBytes(n_rows, nnz), dtype=torch.uint8, device=dev)
    lengths = torch.empty(max(n_rows, 1), dtype=torch.int32, device=dev)
    longest = C.c_int(0)
    ok = lambda status: status == capi.SPGPU_SUCCESS or (_ for _ in ()).throw(RuntimeError(f"status {status}"))
    ok(capi.spgpuCooRowLengthsDevice(handle, _dp(lengths), C.byref(longest), n_rows, nnz, _dp(coo_rows), coo_base, _dp(work)))
    r_idx = None
    rows_in = coo_rows
    if order:
        order_work = torch.empty(capi.spgpuOellOrderWorkBytes(n_rows), dtype=torch.uint8, device=dev)
        r_idx = torch.empty(max(n_rows, 1), dtype=torch.int32, device=dev)
        sorted_lengths = torch.empty(max(n_rows, 1), dtype=torch.int32, device=dev)
        if r_idx_given is not None:      # experiments: an order computed elsewhere (any permutation of the rows)
            r_idx.copy_(r_idx_given)
        else:
            order_call = capi.spgpuOellOrderAlignedDevice if aligned else capi.spgpuOellOrderDevice
            ok(order_call(handle, _dp(r_idx), _dp(sorted_lengths), _dp(lengths), n_rows, window, long_rows, _dp(order_work)))
        inverse = torch.empty(max(n_rows, 1), dtype=torch.int32, device=dev)
        rows_in = torch.empty_like(coo_rows)
        ok(capi.spgpuCooPermuteRowsDevice(handle, _dp(rows_in), _dp(coo_rows), nnz, _dp(r_idx), n_rows, coo_base, _dp(inverse)))
        # returns a host scalar, i.e. synchronises the stream: the order scratch can go afterwards
        ok(capi.spgpuCooRowLengthsDevice(handle, _dp(lengths), C.byref(longest), n_rows, nnz, _dp(rows_in), coo_base, _dp(work)))
        del order_work, inverse
    hacks = (n_rows + hack_size - 1) // hack_size
    hack_offsets = torch.empty(max(hacks, 1), dtype=torch.int32, device=dev)
    height = C.c_int(0)
    ok(capi.spgpuHellPlanDevice(handle, C.byref(height), _dp(hack_offsets), hack_size, n_rows, _dp(lengths), _dp(work)))
    slots = hack_size * height.value
    cM = torch.zeros(max(slots, 1), dtype=coo_vals.dtype, device=dev)
    rP = torch.zeros(max(slots, 1), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ok(capi.spgpuCooToHellDevice(handle, _dp(cM), _dp(rP), _dp(hack_offsets), hack_size, hell_base, n_rows, nnz,
                                 _dp(rows_in), _dp(coo_cols), _dp(coo_vals), coo_base, capi.TYPE_CODE[letter], _dp(lengths),
                                 _dp(work)))
    torch.cuda.synchronize()
    return dict(letter=letter, rows=n_rows, hack_size=hack_size, nnz=nnz, slots=slots, cM=cM, rP=rP, hack_offsets=hack_offsets,
                rS=lengths, rIdx=r_idx, base=hell_base, longest=longest.value)


# ---- device residency + SpMV calls -------------------------------------------------

def to_device(a, device="cuda:0"):
    """numpy array -> torch tensor in HBM (the reference's cudaMalloc + cudaMemcpy)."""
    import torch
    if a is None:
        return None
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _dp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class DeviceHell:
    """A HELL matrix resident in HBM; ``spmv`` is one spgpu?hellspmv call."""

    def __init__(self, hell, device="cuda:0", r_idx=None):
        self.letter, self.rows = hell["letter"], hell["rows"]
        self.hack_size, self.base = hell["hack_size"], hell["base"]
        self.cM, self.rP = to_device(hell["values"], device), to_device(hell["indices"], device)
        self.hack_offsets = to_device(hell["hack_offsets"], device)
        self.rS = to_device(hell["row_lengths"], device)
        self.rIdx = to_device(r_idx, device)
        self.nnz = int(np.sum(hell["row_lengths"], dtype=np.int64))

    def spmv(self, handle, z, y, alpha, x, beta, avg_nnz=0):
        L = self.letter
        capi.hellspmv[L](handle, _dp(z), _dp(y), capi.scalar(L, alpha), _dp(self.cM), _dp(self.rP),
                         self.hack_size, _dp(self.hack_offsets), _dp(self.rS), _dp(self.rIdx), avg_nnz,
                         self.rows, _dp(x), capi.scalar(L, beta), self.base)


class DeviceEll:
    """An ELL matrix resident in HBM; ``spmv`` is one spgpu?ellspmv call."""

    def __init__(self, ell, device="cuda:0", r_idx=None, with_row_sizes=True):
        self.letter, self.rows = ell["letter"], ell["rows"]
        self.pitch, self.max_row, self.base = ell["pitch"], ell["max_row"], ell["base"]
        self.cM, self.rP = to_device(ell["values"], device), to_device(ell["indices"], device)
        self.rS = to_device(ell["row_lengths"], device) if with_row_sizes else None
        self.rIdx = to_device(r_idx, device)
        self.nnz = int(np.sum(ell["row_lengths"], dtype=np.int64))

    def spmv(self, handle, z, y, alpha, x, beta, avg_nnz=0):
        L = self.letter
        capi.ellspmv[L](handle, _dp(z), _dp(y), capi.scalar(L, alpha), _dp(self.cM), _dp(self.rP),
                        self.pitch, self.pitch, _dp(self.rS), _dp(self.rIdx), avg_nnz, self.max_row,
                        self.rows, _dp(x), capi.scalar(L, beta), self.base)


class DeviceHdia:
    """An HDIA matrix resident in HBM; ``spmv`` is one spgpu?hdiaspmv call."""

    def __init__(self, hdia, device="cuda:0"):
        self.letter, self.rows, self.cols = hdia["letter"], hdia["rows"], hdia["cols"]
        self.hack_size = hdia["hack_size"]
        self.dM, self.offsets = to_device(hdia["values"], device), to_device(hdia["offsets"], device)
        self.hack_offsets = to_device(hdia["hack_offsets"], device)

    def spmv(self, handle, z, y, alpha, x, beta):
        L = self.letter
        capi.hdiaspmv[L](handle, _dp(z), _dp(y), capi.scalar(L, alpha), _dp(self.dM), _dp(self.offsets),
                         self.hack_size, _dp(self.hack_offsets), self.rows, self.cols, _dp(x),
                         capi.scalar(L, beta))


class DeviceDia:
    """A DIA matrix resident in HBM; ``spmv`` is one spgpu?diaspmv call."""

    def __init__(self, dia, device="cuda:0"):
        self.letter, self.rows, self.cols = dia["letter"], dia["rows"], dia["cols"]
        self.pitch, self.diags = dia["pitch"], dia["diags"]
        self.dM, self.offsets = to_device(dia["values"], device), to_device(dia["offsets"], device)

    def spmv(self, handle, z, y, alpha, x, beta):
        L = self.letter
        capi.diaspmv[L](handle, _dp(z), _dp(y), capi.scalar(L, alpha), _dp(self.dM), _dp(self.offsets), self.pitch,
                        self.rows, self.cols, self.diags, _dp(x), capi.scalar(L, beta))
