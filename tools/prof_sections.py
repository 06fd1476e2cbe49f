"""In-kernel cycle shares of the NUTS loop (needs a -DSMCN_PROFILE build: SMCN_LIB=...).
    python tools/build_variant.py prof -DSMCN_PROFILE
    SMCN_LIB=smcnuts_amd/variants/libsmcnuts_prof.so python tools/prof_sections.py [steps] [warmup]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smcnuts_amd import ArmaModel, SMCSampler

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N = int(os.environ.get("PROF_N", "65536"))
smc = SMCSampler(K=W + K, N=N, target=ArmaModel(), step_size=0.01, seed=10, save_history=False)
ctx = smc.samples.ctx
if "PROF_SEGS" in os.environ:      # lane queue: segments per block (0: the launcher's rule)
    ctx.call("smcn_set_lane_segments", int(os.environ["PROF_SEGS"]))
smc.run_fused(upto=W, fuse_max=64)
out = (C.c_uint64 * 16)()
ctx.call("smcn_synchronize")
ctx.call("smcn_debug_profile", out, 1)
ctx.timers(reset=True)
smc.run_fused(upto=W + K, fuse_max=64)
smc.finalise_async()
tm = ctx.timers()
ctx.call("smcn_debug_profile", out, 0)
v = np.array(list(out), dtype=np.float64)
names = ["refill", "leapfrog1", "eval", "leaf tests/first", "park/accept (after 9)", "init block (after 11)", "start-doubling",
         "loop-top", "merge loop", "unwind/top-level", "tree end: take record", "tree end: prefetch + rejoin (after 13)", "tree end: x' stores", "tree end: other stores"]
leaps = smc.leapfrogs[W:].sum()
print(f"nuts launches {int(tm[1])}, {tm[0]:.3f} ms total; leapfrogs {leaps}; {leaps / tm[0] / 1e6:.3f} G leapfrog/s in the kernel")
tot = v[:14].sum()
for n, x in zip(names, v[:14]):
    print(f"  {n:18s} {x/tot*100:6.2f}%   {x:.3e}")
if v[14] > 0:
    waves = min(N, 65536) / 64
    print(f"wave-iterations: total {v[14]:.0f}, mean per wave {v[14] / waves:.1f}, longest wave {v[15]:.0f}; "
          f"lane-leapfrogs per wave-iteration {leaps / v[14]:.1f} of 64; cycles per wave-iteration {tot / v[14]:.0f}")
    for n, x in zip(names, v[:14]):
        if x:
            print(f"  {n:28s} {x / v[14]:8.0f} cycles per wave-iteration")
