"""Build libsmcnuts_hip.so (hipcc, gfx950) in-tree."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "smcn_api.hip")
LIB = os.path.join(HERE, "libsmcnuts_hip.so")


def deps():
    """Everything the library is compiled from: every file under csrc/ and the public header."""
    d = sorted(glob.glob(os.path.join(HERE, "csrc", "*.hpp")) + glob.glob(os.path.join(HERE, "csrc", "*.hip")))
    d.append(os.path.join(os.path.dirname(HERE), "include", "smcnuts_hip.h"))
    d.append(os.path.abspath(__file__))      # the compiler flags live here
    return d


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in deps())


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> smcnuts_amd/libsmcnuts_hip.so (cross-compiles without a GPU)."""
    if not force and not is_stale():
        return LIB
    import fcntl
    with open(LIB + ".lock", "w") as lock:          # several ranks may get here at once: one builds, the others wait
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not is_stale():
            return LIB
        return _compile(verbose)


def _compile(verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -amdgpu-atomic-optimizer-strategy=None: the work-queue atomics of the NUTS kernel are issued
    # per particle group and consumed one tree later; the wave-aggregating optimizer would wait for
    # the result at once (readfirstlane), exposing the atomic's round trip on every claim.
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
           "-shared", "-fPIC", "-o", LIB, SRC, "-ldl"]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    tmp = LIB + ".tmp"
    cmd[cmd.index(LIB)] = tmp
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)                             # atomic: a reader never sees a half-written library
    return LIB


if __name__ == "__main__":
    print(build(force=True))
