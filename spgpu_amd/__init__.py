"""spgpu-amd: MI355X-native drop-in for spGPU's ELL/HELL/HDIA SpMV hot path.

The product is the C-ABI shared library ``spgpu_amd/lib/libspgpu.so`` (HIP
kernels for gfx950 + host C code, sources in ``spgpu_amd/csrc``, headers in
``include/spgpu``).  This Python package is plumbing around it for the test
suite and ``bench.py``: ctypes bindings (:mod:`spgpu_amd.capi`), the host
pipeline the reference's harness runs (:mod:`spgpu_amd.formats`) and synthetic
inputs (:mod:`spgpu_amd.synth`).  There is no CPU fallback: importing
:mod:`spgpu_amd.capi` fails if the library has not been built.
"""

__all__ = ["capi", "formats", "synth"]
