"""CPU: the drop-in boundary.  libspgpu.so loads without a GPU, exports every
function include/spgpu/*.h declares, the ctypes layer binds each of them, and the
product has no link to the oracle."""
import glob
import os
import re
import subprocess

from spgpu_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DECL = re.compile(r"^\s*(?:const\s+__device\s+)?(?:void|int|size_t|float|double\*?|long long|spgpuStatus_t|hipStream_t|hipFloatComplex|hipDoubleComplex)\s+"
                  r"(spgpu\w+|computeEll\w+|cooTo\w+|coo2dia|computeHell\w+|ellTo\w+|getHdia\w+|computeHdia\w+|computeDia\w+|diaTo\w+|oellOrder\w*)\s*\(", re.M)


def declared_functions():
    names = set()
    for header in glob.glob(os.path.join(ROOT, "include", "spgpu", "*.h")):
        with open(header) as f:
            names.update(DECL.findall(f.read()))
    return names


def exported_symbols():
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in out.splitlines() if " T " in line}


def test_every_declared_function_is_exported_and_bound():
    declared = declared_functions()
    assert len(declared) == 198, sorted(declared)  # C entry points; the C++ overloads of mmread.hpp are checked in test_mmread.py
    exported = exported_symbols()
    assert declared <= exported, sorted(declared - exported)
    assert declared <= set(capi.DECLARED), sorted(declared - set(capi.DECLARED))
    for name in sorted(declared):
        assert getattr(capi.lib, name) is not None


def test_expected_abi_names_present():
    for letter in "SDCZ":
        for op in ("hellspmv", "ellspmv", "hdiaspmv", "axpby", "maxpby", "dot", "mdot", "nrm2", "mnrm2"):
            assert f"spgpu{letter}{op}" in capi.DECLARED
    for name in ("spgpuCreate", "spgpuDestroy", "spgpuStreamCreate", "spgpuStreamDestroy", "spgpuSetStream",
                 "spgpuGetStream", "spgpuSizeOf", "computeEllRowLenghts", "computeEllAllocPitch", "cooToEll",
                 "computeHellAllocSize", "ellToHell", "getHdiaHacksCount", "computeHdiaHackOffsetsFromCoo", "cooToHdia"):
        assert name in capi.DECLARED


def test_size_of_and_constants():
    assert [capi.spgpuSizeOf(c) for c in range(5)] == [4, 4, 8, 8, 16]
    assert capi.spgpuSizeOf(5) == 0 and capi.spgpuSizeOf(-1) == 0
    with open(os.path.join(ROOT, "include", "spgpu", "hell.h")) as f:
        assert "#define HELL_PITCH_ALIGN_BYTE 128" in f.read()
    with open(os.path.join(ROOT, "include", "spgpu", "ell.h")) as f:
        assert "#define ELL_PITCH_ALIGN_BYTE 128" in f.read()


def test_product_is_gfx950_only_and_oracle_free():
    out = subprocess.run(["nm", "-D", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "orc_" not in out, "product library references oracle symbols"
    deps = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "liboracle" not in deps and "spgpu_ref" not in deps
    assert "libamdhip64" in deps
    # no Python module of the package imports the oracle
    for path in glob.glob(os.path.join(ROOT, "spgpu_amd", "*.py")):
        with open(path) as f:
            src = f.read()
        assert "oracle_api" not in src and "liboracle" not in src, path


def test_headers_compile_as_plain_c(tmp_path):
    src = tmp_path / "abi.c"
    src.write_text('#include "spgpu/core.h"\n#include "spgpu/ell.h"\n#include "spgpu/hell.h"\n#include "spgpu/hdia.h"\n'
                   '#include "spgpu/vector.h"\n#include "spgpu/ell_conv.h"\n#include "spgpu/hell_conv.h"\n'
                   '#include "spgpu/hdia_conv.h"\n'
                   "int main(void){ spgpuHandle_t h = 0; (void)h; return (int)spgpuSizeOf(SPGPU_TYPE_INT) - 4; }\n")
    rocm = "/opt/rocm"
    cmd = ["gcc", "-std=c99", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include", f"-I{ROOT}/include", str(src), "-o",
           str(tmp_path / "abi"), f"-L{os.path.dirname(capi.LIB_PATH)}", "-lspgpu",
           f"-Wl,-rpath,{os.path.dirname(capi.LIB_PATH)}", f"-Wl,-rpath,{rocm}/lib"]
    subprocess.run(cmd, check=True, capture_output=True)
    assert subprocess.run([str(tmp_path / "abi")]).returncode == 0
