"""Same-node context numbers from the vendor library: rocSPARSE's CSR (adaptive, with its analysis step) and ELL SpMV on the
matrices bench.py times -- the reference's own idea of context is its cusparsePerf harness (src/tests/cusparsePerf.cpp:587-776:
cuSPARSE CSR / HYB / ELL beside its kernels).  Loaded with ctypes from /opt/rocm at run time, used ONLY by bench.py's untimed
extras and never by the library: libspgpu.so has no dependency on rocSPARSE (tests/test_capi_surface.py).  Every entry point
returns None when the library, a symbol or a call is not available -- the bench record then carries no vendor number."""
import ctypes as C
import os

_lib = None
_handle = None
OP_NONE = 111          # rocsparse_operation_none
INDEX_BASE_ZERO = 0


def _load():
    global _lib, _handle
    if _lib is not None:
        return _lib or None
    for name in ("librocsparse.so", "/opt/rocm/lib/librocsparse.so", "librocsparse.so.1"):
        try:
            _lib = C.CDLL(name)
            break
        except OSError:
            _lib = False
    if not _lib:
        return None
    try:
        h = C.c_void_p()
        if _lib.rocsparse_create_handle(C.byref(h)) != 0:
            _lib = False
            return None
        _handle = h
    except AttributeError:
        _lib = False
        return None
    return _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _time(stream, fn, reps):
    import torch
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        for _ in range(2):
            if fn() != 0:
                return None
        a.record(stream)
        for _ in range(reps):
            fn()
        b.record(stream)
    b.synchronize()
    return a.elapsed_time(b) / reps


def csrmv_ms(stream, rows, cols, row_ptr, col_ind, val, x, y, letter="D", reps=10):
    """rocsparse_?csrmv with rocsparse_?csrmv_analysis (the adaptive kernel): (ms per SpMV, analysis ms), or None.
    row_ptr int32 [rows + 1], col_ind int32, val: torch tensors on the device; zero-based."""
    import time
    import torch
    lib = _load()
    if lib is None or letter not in "SD":
        return None
    try:
        p = "s" if letter == "S" else "d"
        scalar = C.c_float if letter == "S" else C.c_double
        analysis, mv = getattr(lib, f"rocsparse_{p}csrmv_analysis"), getattr(lib, f"rocsparse_{p}csrmv")
        descr, info = C.c_void_p(), C.c_void_p()
        if lib.rocsparse_create_mat_descr(C.byref(descr)) != 0 or lib.rocsparse_create_mat_info(C.byref(info)) != 0:
            return None
        lib.rocsparse_set_stream(_handle, C.c_void_p(stream.cuda_stream))
        nnz = int(val.numel())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        status = analysis(_handle, OP_NONE, C.c_int(rows), C.c_int(cols), C.c_int(nnz), descr, _ptr(val), _ptr(row_ptr), _ptr(col_ind), info)
        torch.cuda.synchronize()
        analysis_ms = (time.perf_counter() - t0) * 1e3
        if status != 0:
            return None
        alpha, beta = scalar(1.0), scalar(0.0)
        call = lambda: mv(_handle, OP_NONE, C.c_int(rows), C.c_int(cols), C.c_int(nnz), C.byref(alpha), descr, _ptr(val), _ptr(row_ptr),
                          _ptr(col_ind), info, _ptr(x), C.byref(beta), _ptr(y))
        ms = _time(stream, call, reps)
        lib.rocsparse_destroy_mat_info(info)
        lib.rocsparse_destroy_mat_descr(descr)
        return None if ms is None else (ms, analysis_ms)
    except (AttributeError, OSError):
        return None


def ellmv_ms(stream, rows, cols, ell_val, ell_col, width, x, y, letter="D", reps=10):
    """rocsparse_?ellmv on column-major ELL arrays (slot (i, p) at p * rows + i, padding column -1): ms per SpMV, or None."""
    lib = _load()
    if lib is None or letter not in "SD":
        return None
    try:
        p = "s" if letter == "S" else "d"
        scalar = C.c_float if letter == "S" else C.c_double
        mv = getattr(lib, f"rocsparse_{p}ellmv")
        descr = C.c_void_p()
        if lib.rocsparse_create_mat_descr(C.byref(descr)) != 0:
            return None
        lib.rocsparse_set_stream(_handle, C.c_void_p(stream.cuda_stream))
        alpha, beta = scalar(1.0), scalar(0.0)
        call = lambda: mv(_handle, OP_NONE, C.c_int(rows), C.c_int(cols), C.byref(alpha), descr, _ptr(ell_val), _ptr(ell_col), C.c_int(width),
                          _ptr(x), C.byref(beta), _ptr(y))
        ms = _time(stream, call, reps)
        lib.rocsparse_destroy_mat_descr(descr)
        return ms
    except (AttributeError, OSError):
        return None


def uniform_hell_context(stream, h, x, y, reps=10):
    """CSR and ELL of a uniform device HELL dict (synth.hell_uniform_on_device: a dense [hacks][L][32] block) -> dict of ms."""
    import torch
    n, L, letter = h["rows"], h["row_len"], h["letter"]
    out = {}
    hacks = n // 32
    val = h["cM"].view(hacks, L, 32).permute(0, 2, 1).reshape(-1).contiguous()
    col = h["rP"].view(hacks, L, 32).permute(0, 2, 1).reshape(-1).contiguous()
    row_ptr = (torch.arange(n + 1, device=val.device, dtype=torch.int64) * L).to(torch.int32)
    got = csrmv_ms(stream, n, h["cols"], row_ptr, col, val, x, y, letter, reps)
    if got:
        out["rocsparse_csrmv_adaptive_ms"], out["rocsparse_csrmv_analysis_ms"] = round(got[0], 4), round(got[1], 1)
    del val, col, row_ptr
    torch.cuda.empty_cache()
    val = h["cM"].view(hacks, L, 32).permute(1, 0, 2).reshape(-1).contiguous()
    col = h["rP"].view(hacks, L, 32).permute(1, 0, 2).reshape(-1).contiguous()
    ms = ellmv_ms(stream, n, h["cols"], val, col, L, x, y, letter, reps)
    if ms:
        out["rocsparse_ellmv_ms"] = round(ms, 4)
    del val, col
    torch.cuda.empty_cache()
    return out or None


def coo_context(stream, n_rows, n_cols, lengths, coo_cols, coo_vals, x, y, letter="D", reps=10):
    """CSR of row-major COO triplets (synth.ragged_coo_on_device) with the given row lengths -> dict of ms."""
    import numpy as np
    import torch
    row_ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(np.asarray(lengths, np.int64))]).astype(np.int32)).to(coo_cols.device)
    got = csrmv_ms(stream, n_rows, n_cols, row_ptr, coo_cols, coo_vals, x, y, letter, reps)
    if not got:
        return None
    return dict(rocsparse_csrmv_adaptive_ms=round(got[0], 4), rocsparse_csrmv_analysis_ms=round(got[1], 1))
