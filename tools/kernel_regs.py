#!/usr/bin/env python3
"""Registers / scratch / LDS of the gfx950 kernels in an object file or shared library of this build (no GPU needed):
    python tools/kernel_regs.py big-linear-algebra_amd/csrc/bla_gemm.o [name-substring ...]
Reads the code object's metadata notes (.vgpr_count, .agpr_count, .private_segment_fixed_size = scratch bytes per lane, .group_segment_fixed_size)."""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
path = sys.argv[1]; pats = sys.argv[2:]
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat"), os.path.join(d, "co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"])
    notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co], text=True)
for blk in notes.split("- .agpr_count:")[1:]:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk)
    name = g("name").group(1)
    try:
        name = subprocess.check_output([LLVM + "/llvm-cxxfilt", name], text=True).strip()
    except Exception:
        pass
    if pats and not all(p in name for p in pats):
        continue
    print(f"agpr {blk.split()[0]:>3} vgpr {g('vgpr_count').group(1):>3} sgpr {g('sgpr_count').group(1):>3} scratch {g('private_segment_fixed_size').group(1):>5} lds {g('group_segment_fixed_size').group(1):>6}  {name[:200]}")
