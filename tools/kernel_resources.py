#!/usr/bin/env python3
"""Build-container helper: registers / LDS / scratch of every kernel in one .hip file (device-only compile for gfx950,
llvm-readelf notes).  usage: tools/kernel_resources.py spgpu_amd/csrc/ellpack_spmv.hip [substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    src, wanted = sys.argv[1], sys.argv[2:]
    extra = os.environ.get("EXTRA_HIPFLAGS", "").split()
    with tempfile.TemporaryDirectory() as tmp:
        bundle, elf = os.path.join(tmp, "k.co"), os.path.join(tmp, "k.elf")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include",
                        f"-I{ROOT}/spgpu_amd/csrc", "--cuda-device-only", "-c", src, "-o", bundle] + extra, check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={bundle}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={elf}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", elf], capture_output=True, text=True, check=True).stdout
        if os.environ.get("KEEP_ELF"):
            os.replace(elf, os.environ["KEEP_ELF"])
    print(f"{'vgpr':>5} {'sgpr':>5} {'lds':>7} {'scratch':>7}  kernel")
    for block in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        get = lambda key: re.search(rf"\.{key}:\s+(\S+)", block).group(1)
        name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
        name = name.replace("spgpu::", "")
        if wanted and not all(w in name for w in wanted):
            continue
        print(f"{get('vgpr_count'):>5} {get('sgpr_count'):>5} {get('group_segment_fixed_size'):>7} "
              f"{get('private_segment_fixed_size'):>7}  {name[:170]}")


if __name__ == "__main__":
    main()
