"""config 4 with EVERY tree in the wave-per-tree kernel (smcn_set_nuts_cap(0, 2)) against the shipped two-phase launch."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from smcnuts_amd import PRMwCDModel, SMCSampler
for label, cap in (("two-phase (9, 1)", None), ("finisher kernel for every tree (0, 2)", (0, 2))):
    smc = SMCSampler(K=22, N=65536, target=PRMwCDModel(), step_size=0.01, lkernel="GaussianApproxLKernel", tempering=True, seed=10,
                     save_history=False)
    if cap is not None:
        smc.samples.ctx.call("smcn_set_nuts_cap", cap[0], cap[1])
    for _ in range(12):
        smc.step()
    smc.samples.ctx.call("smcn_synchronize")
    smc.samples.ctx.timers(reset=True)
    t0 = time.perf_counter()
    for _ in range(10):
        smc.step()
    smc.samples.ctx.call("smcn_synchronize")
    dt = time.perf_counter() - t0
    tm = smc.samples.ctx.timers()
    lf = int(smc.leapfrogs[12:].sum())
    print(f"{label}: {lf / dt / 1e9:.3f} G leapfrog/s, {1e3 * dt / 10:.2f} ms per step, NUTS launches {tm[0] / 10:.2f} ms per step")
