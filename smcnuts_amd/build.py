"""Build libsmcnuts_hip.so (hipcc, gfx950) in-tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "smcn_api.hip")
LIB = os.path.join(HERE, "libsmcnuts_hip.so")
DEPS = [os.path.join(HERE, "csrc", f) for f in
        ("smcn_api.hip", "smcn_nuts.hpp", "smcn_models.hpp", "smcn_weights.hpp", "smcn_device.hpp", "smcn_step.hpp", "smcn_nuts2.hpp")]
DEPS.append(os.path.join(os.path.dirname(HERE), "include", "smcnuts_hip.h"))


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> smcnuts_amd/libsmcnuts_hip.so (cross-compiles without a GPU)."""
    if not force and not is_stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -amdgpu-atomic-optimizer-strategy=None: the work-queue atomics of the NUTS kernel are issued
    # per particle group and consumed one tree later; the wave-aggregating optimizer would wait for
    # the result at once (readfirstlane), exposing the atomic's round trip on every claim.
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
           "-shared", "-fPIC", "-o", LIB, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
