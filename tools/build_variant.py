"""Diagnostic / A-B builds of the library (same ABI), kept out of the product .so:
    python tools/build_variant.py <tag> [-DNAME ...]   ->  smcnuts_amd/variants/libsmcnuts_<tag>.so
Load one with SMCN_LIB=<path>."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, defs = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "smcnuts_amd", "variants", f"libsmcnuts_{tag}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
       "-shared", "-fPIC", *defs, "-o", out, os.path.join(ROOT, "smcnuts_amd", "csrc", "smcn_api.hip"), "-ldl"]
subprocess.check_call(cmd)
print(out)
