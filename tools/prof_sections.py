"""In-kernel cycle shares of the NUTS loop (needs a -DSMCN_PROFILE build: SMCN_LIB=...)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smcnuts_amd import ArmaModel, SMCSampler

smc = SMCSampler(K=20, N=65536, target=ArmaModel(), step_size=0.01, seed=10, save_history=False)
for k in range(10):
    smc.step()
ctx = smc.samples.ctx
out = (C.c_uint64 * 8)()
ctx.call("smcn_debug_profile", out, 1)
ctx.timers(reset=True)
for k in range(10):
    smc.step()
tm = ctx.timers()
ctx.call("smcn_debug_profile", out, 0)
v = np.array(list(out), dtype=np.float64)
names = ["fetch+refill", "pre(leapfrog1)", "eval", "post-leaf", "merges", "end-doubling/emit", "start-doubling", "loop-top(init)"]
print("max resident blocks (census):", int(v[6]))
v[6] = 0
tot = v.sum()
print(f"nuts avg launch {tm[0]/tm[1]:.3f} ms; leapfrogs/launch {smc.leapfrogs[10:].mean():.0f}; total wave-cycles {tot:.3e}")
for n, x in zip(names, v):
    print(f"  {n:16s} {x/tot*100:6.2f}%   {x:.3e}")
