import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()
    try:
        config._lab_build = lab_build()
    except Exception:  # noqa: BLE001 - no library: the tests that need it fail on their own
        config._lab_build = False


def _ensure_built():
    """The built artefacts are git-ignored and normally travel with the tree; if a checkout arrives without
    them, build them once (same toolchain on every box: hipcc for the library, gcc for oracle and tools).
    oracle/_ref needs /root/reference and is only ever built in the build container."""
    import subprocess
    need = {"lib": os.path.join(ROOT, "spgpu_amd", "lib", "libspgpu.so"),
            "oracle": os.path.join(ROOT, "oracle", "liboracle.so"),
            "tools": os.path.join(ROOT, "tools", "cg_amd.bin")}
    missing = [target for target, path in need.items() if not os.path.exists(path)]
    if missing:
        subprocess.run(["make", "-C", ROOT, *missing], check=False)


def lab_build():
    """True if libspgpu.so carries the non-default kernel shapes (-DSPGPU_TUNING_VARIANTS, include/spgpu/tuning.h)."""
    from spgpu_amd import capi
    return bool(capi.spgpuTuningVariantsBuilt())


# a test of a non-default kernel shape: runs on a library built with EXTRA_HIPFLAGS=-DSPGPU_TUNING_VARIANTS (the A/B tools'
# build), skipped on the product build, which ignores the knobs that select those shapes
lab = pytest.mark.skipif("not config._lab_build", reason="non-default kernel shape: needs a -DSPGPU_TUNING_VARIANTS build")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def gpu():
    """Handle + torch device for the -m gpu tests; fails loudly without the native library."""
    import torch
    from spgpu_amd import capi  # raises if libspgpu.so is missing
    assert torch.cuda.is_available(), "gpu-marked test started without a GPU"
    handle = capi.create_handle(0)
    yield handle
    torch.cuda.synchronize()
    capi.spgpuDestroy(handle)


@pytest.fixture
def tuning(monkeypatch):
    """Set tuning knobs of the library (include/spgpu/tuning.h) for one test: tuning(SPGPU_SPMV_VARIANT=3).
    The library caches them, so they are reloaded here and again after the environment is restored."""
    from spgpu_amd import capi

    def set_knobs(**knobs):
        for name, value in knobs.items():
            monkeypatch.setenv(name, str(value))
        capi.spgpuTuningReload()

    yield set_knobs
    monkeypatch.undo()
    capi.spgpuTuningReload()
