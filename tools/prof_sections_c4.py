"""In-kernel cycle shares of the group NUTS kernel on config 4 (PRMwCD) or config 5 (needs a -DSMCN_PROFILE build):
    python tools/build_variant.py prof -DSMCN_PROFILE
    SMCN_LIB=smcnuts_amd/variants/libsmcnuts_prof.so python tools/prof_sections_c4.py [iterations] [warmup]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smcnuts_amd import IsoGaussian, PRMwCDModel, SMCSampler

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W = int(sys.argv[2]) if len(sys.argv) > 2 else 12
CFG = sys.argv[3] if len(sys.argv) > 3 else "c4"      # "c5" [step size]: iso-Gaussian D = 256, N = 131072
if CFG == "c5":
    smc = SMCSampler(K=W + K, N=131072, target=IsoGaussian(256), step_size=float(sys.argv[4]) if len(sys.argv) > 4 else 0.25,
                     seed=10, save_history=False)
else:
    smc = SMCSampler(K=W + K, N=65536, target=PRMwCDModel(), step_size=0.01, lkernel="GaussianApproxLKernel", tempering=True,
                     seed=10, save_history=False)
for _ in range(W):
    smc.step()
ctx = smc.samples.ctx
out = (C.c_uint64 * 16)()
ctx.call("smcn_synchronize")
ctx.call("smcn_debug_profile", out, 1)
ctx.timers(reset=True)
for _ in range(K):
    smc.step()
tm = ctx.timers()
ctx.call("smcn_debug_profile", out, 0)
v = np.array(list(out), dtype=np.float64)
# PROF(s) closes the section that ends at stamp s (smcn_nuts.hpp)
names = ["fetch work / loop top", "leapfrog, first half", "eval (value + gradient)", "leaf: second half, tests, first-leaf store",
         "merge loop", "end of doubling / tree end", "-", "loop back-edge"]
leaps = smc.leapfrogs[W:].sum()
print(f"nuts launches {int(tm[1])}, {tm[0]:.3f} ms total; leapfrogs {leaps}; {leaps / tm[0] / 1e6:.3f} G leapfrog/s in the kernel")
tot = v[:14].sum()
for n, x in zip(names, v[:8]):
    print(f"  {n:44s} {x / tot * 100:6.2f}%   {x:.3e}")
