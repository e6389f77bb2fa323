"""ctypes bindings of the spgpu-amd C ABI (``include/spgpu/*.h``).

Every entry point is bound with its exact C signature; nothing here computes.
Device arrays are passed as raw addresses (``tensor.data_ptr()``), scalars by
value, exactly like a C caller of the reference would (``hell.h:45-59``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPGPU_LIB") or os.path.join(_HERE, "lib", "libspgpu.so")  # SPGPU_LIB: A/B of two builds


class MissingNativeLibrary(ImportError):
    pass


if not os.path.exists(LIB_PATH):
    raise MissingNativeLibrary(
        f"{LIB_PATH} not found: build it with `make lib` (or __graft_entry__.build()); "
        "spgpu-amd has no Python or CPU fallback for its kernels")

# torch (device-memory plumbing of the tests and bench.py) ships its own HIP runtime with the
# same soname as /opt/rocm's; whichever is loaded first serves the whole process.  Load torch's
# first so that the tensors it allocates and the kernels launched here share ONE runtime.
try:
    import torch  # noqa: F401
except ImportError:  # a C-only deployment links /opt/rocm's runtime directly
    pass

lib = C.CDLL(LIB_PATH)

# ---- types ------------------------------------------------------------------
SPGPU_SUCCESS, SPGPU_UNSUPPORTED, SPGPU_UNSPECIFIED, SPGPU_OUTOFMEMORY = 0, 1, 2, 3
TYPE_INT, TYPE_FLOAT, TYPE_DOUBLE, TYPE_COMPLEX_FLOAT, TYPE_COMPLEX_DOUBLE = range(5)


class FloatComplex(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class DoubleComplex(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double)]


class HandleStruct(C.Structure):
    """Public fields of SpgpuHandleStruct (include/spgpu/core.h; reference core.h:60-82)."""
    _fields_ = [
        ("currentStream", C.c_void_p), ("defaultStream", C.c_void_p),
        ("device", C.c_int), ("warpSize", C.c_int), ("maxThreadsPerBlock", C.c_int),
        ("maxGridSizeX", C.c_int), ("maxGridSizeY", C.c_int), ("maxGridSizeZ", C.c_int),
        ("multiProcessorCount", C.c_int), ("capabilityMajor", C.c_int), ("capabilityMinor", C.c_int),
    ]


Handle = C.POINTER(HandleStruct)
ptr = C.c_void_p  # device or host address
i32 = C.c_int

# scalar C type per letter of the API
SCALAR = {"S": C.c_float, "D": C.c_double, "C": FloatComplex, "Z": DoubleComplex}
REAL = {"S": C.c_float, "D": C.c_double, "C": C.c_float, "Z": C.c_double}
TYPE_CODE = {"S": TYPE_FLOAT, "D": TYPE_DOUBLE, "C": TYPE_COMPLEX_FLOAT, "Z": TYPE_COMPLEX_DOUBLE}


def scalar(letter, value):
    """Python number -> the by-value C scalar of the given API flavour."""
    if letter in "SD":
        return SCALAR[letter](float(value))
    v = complex(value)
    return SCALAR[letter](v.real, v.imag)


def _bind(name, restype, argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


DECLARED = {}  # name -> (restype, argtypes); the test suite checks it against the headers


def _decl(name, restype, argtypes):
    DECLARED[name] = (restype, argtypes)
    return _bind(name, restype, argtypes)


# ---- core.h -------------------------------------------------------------------
spgpuCreate = _decl("spgpuCreate", i32, [C.POINTER(Handle), i32])
spgpuDestroy = _decl("spgpuDestroy", None, [Handle])
spgpuStreamCreate = _decl("spgpuStreamCreate", None, [Handle, C.POINTER(C.c_void_p)])
spgpuStreamDestroy = _decl("spgpuStreamDestroy", None, [C.c_void_p])
spgpuSetStream = _decl("spgpuSetStream", None, [Handle, C.c_void_p])
spgpuGetStream = _decl("spgpuGetStream", C.c_void_p, [Handle])
spgpuSizeOf = _decl("spgpuSizeOf", C.c_size_t, [i32])

# ---- hell.h / ell.h / hdia.h ----------------------------------------------------
hellspmv, ellspmv, hdiaspmv = {}, {}, {}
axpby, maxpby, dot, mdot, nrm2, mnrm2 = {}, {}, {}, {}, {}, {}
for _L, _T in SCALAR.items():
    hellspmv[_L] = _decl(f"spgpu{_L}hellspmv", None,
                         [Handle, ptr, ptr, _T, ptr, ptr, i32, ptr, ptr, ptr, i32, i32, ptr, _T, i32])
    ellspmv[_L] = _decl(f"spgpu{_L}ellspmv", None,
                        [Handle, ptr, ptr, _T, ptr, ptr, i32, i32, ptr, ptr, i32, i32, i32, ptr, _T, i32])
    hdiaspmv[_L] = _decl(f"spgpu{_L}hdiaspmv", None,
                         [Handle, ptr, ptr, _T, ptr, ptr, i32, ptr, i32, i32, ptr, _T])
    # ---- vector.h ---------------------------------------------------------------
    axpby[_L] = _decl(f"spgpu{_L}axpby", None, [Handle, ptr, i32, _T, ptr, _T, ptr])
    maxpby[_L] = _decl(f"spgpu{_L}maxpby", None, [Handle, ptr, i32, _T, ptr, _T, ptr, i32, i32])
    dot[_L] = _decl(f"spgpu{_L}dot", _T, [Handle, i32, ptr, ptr])
    mdot[_L] = _decl(f"spgpu{_L}mdot", None, [Handle, ptr, i32, ptr, ptr, i32, i32])
    nrm2[_L] = _decl(f"spgpu{_L}nrm2", REAL[_L], [Handle, i32, ptr])
    mnrm2[_L] = _decl(f"spgpu{_L}mnrm2", None, [Handle, ptr, i32, ptr, i32, i32])

# ---- vector.h, second half: the rest of the reference's Level-1 ----------------------
scal, vabs, axy, maxy, axypbz, maxypbz, gath, scat, setscal = {}, {}, {}, {}, {}, {}, {}, {}, {}
asum, amax, masum, mamax = {}, {}, {}, {}
for _L, _T in list(SCALAR.items()) + [("I", C.c_int)]:
    gath[_L] = _decl(f"spgpu{_L}gath", None, [Handle, ptr, i32, ptr, i32, ptr])
    scat[_L] = _decl(f"spgpu{_L}scat", None, [Handle, ptr, i32, ptr, ptr, i32, _T])
    setscal[_L] = _decl(f"spgpu{_L}setscal", None, [Handle, i32, i32, i32, _T, ptr])
    if _L == "I":
        continue
    scal[_L] = _decl(f"spgpu{_L}scal", None, [Handle, ptr, i32, _T, ptr])
    vabs[_L] = _decl(f"spgpu{_L}abs", None, [Handle, ptr, i32, _T, ptr])
    axy[_L] = _decl(f"spgpu{_L}axy", None, [Handle, ptr, i32, _T, ptr, ptr])
    maxy[_L] = _decl(f"spgpu{_L}maxy", None, [Handle, ptr, i32, _T, ptr, ptr, i32, i32])
    axypbz[_L] = _decl(f"spgpu{_L}axypbz", None, [Handle, ptr, i32, _T, ptr, _T, ptr, ptr])
    maxypbz[_L] = _decl(f"spgpu{_L}maxypbz", None, [Handle, ptr, i32, _T, ptr, _T, ptr, ptr, i32, i32])
    asum[_L] = _decl(f"spgpu{_L}asum", REAL[_L], [Handle, i32, ptr])
    amax[_L] = _decl(f"spgpu{_L}amax", REAL[_L], [Handle, i32, ptr])
    masum[_L] = _decl(f"spgpu{_L}masum", None, [Handle, ptr, i32, ptr, i32, i32])
    mamax[_L] = _decl(f"spgpu{_L}mamax", None, [Handle, ptr, i32, ptr, i32, i32])

# ---- spmm.h (new: multi-vector product of the row-sharded path) ---------------------
hellspmm, mv_interleave, mv_deinterleave = {}, {}, {}
for _L in "SD":
    _T = SCALAR[_L]
    hellspmm[_L] = _decl(f"spgpu{_L}hellspmm", None,
                         [Handle, ptr, ptr, _T, ptr, ptr, i32, ptr, ptr, ptr, i32, i32, ptr, _T, i32, i32, i32, i32])
    mv_interleave[_L] = _decl(f"spgpu{_L}mvInterleave", None, [Handle, ptr, i32, ptr, i32, i32, i32])
    mv_deinterleave[_L] = _decl(f"spgpu{_L}mvDeinterleave", None, [Handle, ptr, i32, ptr, i32, i32, i32])

# ---- ell_conv.h / hell_conv.h / hdia_conv.h (host pointers) --------------------------
computeEllRowLenghts = _decl("computeEllRowLenghts", None, [ptr, C.POINTER(i32), i32, i32, ptr, i32])
computeEllAllocPitch = _decl("computeEllAllocPitch", i32, [i32])
cooToEll = _decl("cooToEll", None, [ptr, ptr, i32, i32, i32, i32, i32, i32, ptr, ptr, ptr, i32, i32])
computeHellAllocSize = _decl("computeHellAllocSize", None, [C.POINTER(i32), i32, i32, ptr])
ellToHell = _decl("ellToHell", None, [ptr, ptr, ptr, i32, ptr, ptr, i32, i32, ptr, i32, i32])
getHdiaHacksCount = _decl("getHdiaHacksCount", i32, [i32, i32])
computeHdiaHackOffsetsFromCoo = _decl("computeHdiaHackOffsetsFromCoo", None,
                                      [C.POINTER(i32), ptr, i32, i32, i32, i32, ptr, ptr, i32])
cooToHdia = _decl("cooToHdia", None, [ptr, ptr, ptr, i32, i32, i32, i32, ptr, ptr, ptr, i32, i32])


# ---- dia.h / dia_conv.h / DIA->HDIA / OELL / csput ---------------------------------
diaspmv, ellcsput = {}, {}
for _L, _T in SCALAR.items():
    diaspmv[_L] = _decl(f"spgpu{_L}diaspmv", None, [Handle, ptr, ptr, _T, ptr, ptr, i32, i32, i32, i32, ptr, _T])
    ellcsput[_L] = _decl(f"spgpu{_L}ellcsput", None, [Handle, _T, ptr, ptr, i32, i32, ptr, i32, ptr, ptr, ptr, i32])
computeDiaAllocPitch = _decl("computeDiaAllocPitch", i32, [i32])
computeDiaDiagonalsCount = _decl("computeDiaDiagonalsCount", i32, [i32, i32, i32, ptr, ptr])
coo2dia = _decl("coo2dia", None, [ptr, ptr, i32, i32, i32, i32, i32, ptr, ptr, ptr, i32, i32])
computeHdiaHackOffsets = _decl("computeHdiaHackOffsets", None, [C.POINTER(i32), ptr, i32, ptr, i32, i32, i32, i32])
diaToHdia = _decl("diaToHdia", None, [ptr, ptr, ptr, i32, ptr, ptr, i32, i32, i32, i32])
ellToOell = _decl("ellToOell", None, [ptr, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, i32, i32])


# ---- convert_device.h (new: COO -> ELL / HELL in HBM) -----------------------------------------------
spgpuCooConvertWorkBytes = _decl("spgpuCooConvertWorkBytes", C.c_size_t, [i32, i32])
spgpuCooRowLengthsDevice = _decl("spgpuCooRowLengthsDevice", i32, [Handle, ptr, C.POINTER(i32), i32, i32, ptr, i32, ptr])
spgpuCooToEllDevice = _decl("spgpuCooToEllDevice", i32, [Handle, ptr, ptr, i32, i32, i32, i32, i32, ptr, ptr, ptr, i32, i32, ptr, ptr])
spgpuHellPlanDevice = _decl("spgpuHellPlanDevice", i32, [Handle, C.POINTER(i32), ptr, i32, i32, ptr, ptr])
spgpuCooToHellDevice = _decl("spgpuCooToHellDevice", i32, [Handle, ptr, ptr, ptr, i32, i32, i32, i32, ptr, ptr, ptr, i32, i32, ptr, ptr])
spgpuCooDiaWorkBytes = _decl("spgpuCooDiaWorkBytes", C.c_size_t, [i32, i32])
spgpuCooDiaPlanDevice = _decl("spgpuCooDiaPlanDevice", i32, [Handle, C.POINTER(i32), i32, i32, i32, ptr, ptr, i32, ptr])
spgpuCooToDiaScratchBytes = _decl("spgpuCooToDiaScratchBytes", C.c_size_t, [i32, i32])
spgpuCooToDiaDevice = _decl("spgpuCooToDiaDevice", i32, [Handle, ptr, ptr, i32, i32, i32, i32, i32, ptr, ptr, ptr, i32, i32, ptr, ptr])
spgpuCooHdiaPlanWorkBytes = _decl("spgpuCooHdiaPlanWorkBytes", C.c_size_t, [i32, i32])
spgpuCooHdiaPlanDevice = _decl("spgpuCooHdiaPlanDevice", i32, [Handle, C.POINTER(i32), ptr, i32, i32, i32, i32, ptr, ptr, i32, ptr])
spgpuCooToHdiaScratchBytes = _decl("spgpuCooToHdiaScratchBytes", C.c_size_t, [i32, i32])
spgpuCooToHdiaDevice = _decl("spgpuCooToHdiaDevice", i32, [Handle, ptr, ptr, ptr, i32, i32, i32, i32, ptr, ptr, ptr, i32, i32, i32, ptr, ptr])


# ---- oell_device.h (new: rows ordered by length in HBM) + the host order (ell_conv.h) ----------------------------
oellOrder = _decl("oellOrder", None, [ptr, ptr, ptr, i32, i32, i32])
oellOrderAligned = _decl("oellOrderAligned", None, [ptr, ptr, ptr, i32, i32, i32])
spgpuOellOrderWorkBytes = _decl("spgpuOellOrderWorkBytes", C.c_size_t, [i32])
spgpuOellOrderDevice = _decl("spgpuOellOrderDevice", i32, [Handle, ptr, ptr, ptr, i32, i32, i32, ptr])
spgpuOellOrderAlignedDevice = _decl("spgpuOellOrderAlignedDevice", i32, [Handle, ptr, ptr, ptr, i32, i32, i32, ptr])
spgpuEllToOellDevice = _decl("spgpuEllToOellDevice", i32, [Handle, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, i32, i32, i32, i32, ptr])
spgpuCooPermuteRowsDevice = _decl("spgpuCooPermuteRowsDevice", i32, [Handle, ptr, ptr, i32, ptr, i32, i32, ptr])


# ---- sharded.h (new: the row-sharded SpMM driver in C, RCCL through dlopen) ----------------------------------------
class HellBlockD(C.Structure):
    """spgpuHellBlockD: one HELL row block in device memory."""
    _fields_ = [("cM", C.c_void_p), ("rP", C.c_void_p), ("hackSize", C.c_int), ("hackOffsets", C.c_void_p), ("rS", C.c_void_p),
                ("rows", C.c_int), ("avgNnzPerRow", C.c_int), ("baseIndex", C.c_int), ("slots", C.c_longlong)]


EXCHANGE_ALLGATHER, EXCHANGE_NEEDED = 0, 1
ShardedPlan = C.c_void_p
spgpuCommAvailable = _decl("spgpuCommAvailable", i32, [])
spgpuCommGetUniqueId = _decl("spgpuCommGetUniqueId", i32, [ptr])
spgpuCommInitRank = _decl("spgpuCommInitRank", i32, [C.POINTER(C.c_void_p), i32, ptr, i32])
spgpuCommInitAll = _decl("spgpuCommInitAll", i32, [C.POINTER(C.c_void_p), i32, ptr])
spgpuCommDestroy = _decl("spgpuCommDestroy", None, [ptr])
spgpuDhellspmmShardedCreate = _decl("spgpuDhellspmmShardedCreate", i32,
                                    [C.POINTER(ShardedPlan), Handle, ptr, i32, i32, C.POINTER(C.c_longlong), C.POINTER(HellBlockD),
                                     C.POINTER(HellBlockD), i32, i32])
spgpuDhellspmmShardedStep = _decl("spgpuDhellspmmShardedStep", i32, [ShardedPlan, ptr, ptr, C.c_double, ptr, C.c_double])
spgpuDhellspmmShardedExchange = _decl("spgpuDhellspmmShardedExchange", i32, [ShardedPlan, ptr])
spgpuDhellspmmShardedExchangeWait = _decl("spgpuDhellspmmShardedExchangeWait", i32, [ShardedPlan])
spgpuDhellspmmShardedProducts = _decl("spgpuDhellspmmShardedProducts", i32, [ShardedPlan, ptr, ptr, C.c_double, ptr, C.c_double])
spgpuDhellspmmShardedRowsReceived = _decl("spgpuDhellspmmShardedRowsReceived", C.c_longlong, [ShardedPlan])
spgpuDhellspmmShardedExchanged = _decl("spgpuDhellspmmShardedExchanged", C.c_void_p, [ShardedPlan, C.POINTER(C.c_longlong)])
spgpuDhellspmmShardedDestroy = _decl("spgpuDhellspmmShardedDestroy", None, [ShardedPlan])


def hell_block(part, avg_nnz=0):
    """A device HELL dict (cM, rP, hack_offsets, rS tensors; rows, hack_size) as the C struct; keep `part` alive."""
    p = lambda t: t.data_ptr()
    slots = int(part["slots"]) if "slots" in part else int(part["cM"].numel())
    return HellBlockD(p(part["cM"]), p(part["rP"]), part["hack_size"], p(part["hack_offsets"]), p(part["rS"]), part["rows"], avg_nnz,
                      part.get("base", 0), slots)


# ---- mmread.h (C wrappers of the Matrix Market reader) --------------------------------------------------------
spgpuMmProperties = _decl("spgpuMmProperties", i32, [C.c_char_p, ptr])
spgpuMmReadCoo = _decl("spgpuMmReadCoo", i32, [C.c_char_p, C.c_char, ptr, ptr, ptr])
spgpuMmUnfoldedSizeD = _decl("spgpuMmUnfoldedSizeD", i32, [ptr, ptr, ptr, i32])
spgpuMmUnfoldD = _decl("spgpuMmUnfoldD", None, [ptr, ptr, ptr, ptr, ptr, ptr, i32])


# ---- device_scalars.h (new: results and coefficients in device memory, graph-capturable) ---------------------
dot_device, axpby_device, axpby_quot_device, div_device, nrm2_device = {}, {}, {}, {}, {}
for _L in "SD":
    dot_device[_L] = _decl(f"spgpu{_L}dotDevice", None, [Handle, ptr, i32, ptr, ptr])
    nrm2_device[_L] = _decl(f"spgpu{_L}nrm2Device", None, [Handle, ptr, i32, ptr])
    axpby_device[_L] = _decl(f"spgpu{_L}axpbyDevice", None, [Handle, ptr, i32, ptr, ptr, ptr, ptr])
    axpby_quot_device[_L] = _decl(f"spgpu{_L}axpbyQuotDevice", None, [Handle, ptr, i32, ptr, ptr, ptr, ptr, ptr, i32, ptr])
    div_device[_L] = _decl(f"spgpu{_L}divDevice", None, [Handle, ptr, ptr, ptr, i32])
hellspmv_dot_device, axpby_pair_dot_device = {}, {}
for _L in "SD":
    _S = SCALAR[_L]
    hellspmv_dot_device[_L] = _decl(f"spgpu{_L}hellspmvDotDevice", None,
                                    [Handle, ptr, ptr, ptr, ptr, _S, ptr, ptr, i32, ptr, ptr, i32, ptr, _S, i32])
    axpby_pair_dot_device[_L] = _decl(f"spgpu{_L}axpbyPairDotDevice", None,
                                      [Handle, ptr, i32, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr])


# ---- tuning.h: per-handle kernel-form hint ---------------------------------------------------------------------
FORM_AUTO, FORM_GATHER, FORM_STRIPS, FORM_XTILE, FORM_SWEEP = range(5)
spgpuSetSpmvForm = _decl("spgpuSetSpmvForm", None, [Handle, i32])
spgpuGetSpmvForm = _decl("spgpuGetSpmvForm", i32, [Handle])
spgpuDeepListOverflows = _decl("spgpuDeepListOverflows", i32, [Handle])
spgpuDeepListFallbacks = _decl("spgpuDeepListFallbacks", i32, [Handle])
spgpuDeepListsRecycled = _decl("spgpuDeepListsRecycled", i32, [Handle])
spgpuSpmvPlanCounts = _decl("spgpuSpmvPlanCounts", None, [Handle, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)])
spgpuHellSpmvPrepare = _decl("spgpuHellSpmvPrepare", i32, [Handle, i32, ptr, ptr, i32, ptr, ptr, ptr, i32, i32])
spgpuHellSpmvFreeze = _decl("spgpuHellSpmvFreeze", i32, [Handle, i32, ptr, ptr, i32, ptr, ptr, ptr, i32, i32])
spgpuEllSpmvFreeze = _decl("spgpuEllSpmvFreeze", i32, [Handle, i32, ptr, ptr, i32, i32, ptr, ptr, i32, i32, i32])
spgpuHellSpmvAdopt = _decl("spgpuHellSpmvAdopt", i32, [Handle, i32, ptr, ptr, i32, ptr, ptr, i32, i32])
spgpuEllSpmvAdopt = _decl("spgpuEllSpmvAdopt", i32, [Handle, i32, ptr, ptr, i32, i32, ptr, i32, i32, i32])
spgpuHellSpmvOptimize = _decl("spgpuHellSpmvOptimize", i32, [Handle, i32, ptr, ptr, i32, ptr, ptr, ptr, i32, i32])
SPMV_AS_IS, SPMV_FROZEN, SPMV_ADOPTED = 0, 1, 2
spgpuSpmvAdoptedUses = _decl("spgpuSpmvAdoptedUses", i32, [Handle])
spgpuSpmvThaw = _decl("spgpuSpmvThaw", i32, [Handle, ptr])
spgpuSpmvFrozenBytes = _decl("spgpuSpmvFrozenBytes", C.c_longlong, [Handle])
spgpuEllSpmvPrepare = _decl("spgpuEllSpmvPrepare", i32, [Handle, i32, ptr, ptr, i32, i32, ptr, ptr, i32, i32, i32])


def plan_counts(handle):
    """(launches with a plan, analyses started, plans found stale) of the handle's ordered ELL/HELL SpMVs (tuning.h)."""
    u, b, s = i32(0), i32(0), i32(0)
    spgpuSpmvPlanCounts(handle, C.byref(u), C.byref(b), C.byref(s))
    return u.value, b.value, s.value
spgpuGetLastSpmvForm = _decl("spgpuGetLastSpmvForm", i32, [Handle])
spgpuTuningVariantsBuilt = _decl("spgpuTuningVariantsBuilt", i32, [])
spgpuHellSpmvForm = _decl("spgpuHellSpmvForm", i32, [Handle, i32, ptr, i32, ptr, ptr, i32, i32])
spgpuEllSpmvForm = _decl("spgpuEllSpmvForm", i32, [Handle, i32, ptr, i32, ptr, i32, i32, i32])

# ---- tuning.h: environment knobs are cached by the library; call after changing one ---------------------------
spgpuTuningReload = _decl("spgpuTuningReload", None, [])


def create_handle(device=0):
    h = Handle()
    status = spgpuCreate(C.byref(h), device)
    if status != SPGPU_SUCCESS or not h:
        raise RuntimeError(f"spgpuCreate(device={device}) failed with status {status}")
    return h
