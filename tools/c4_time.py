import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smcnuts_amd import PRMwCDModel, SMCSampler
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
smc = SMCSampler(K=K, N=N, target=PRMwCDModel(), step_size=0.01, lkernel="GaussianApproxLKernel", tempering=True, seed=5, save_history=False)
for k in range(K):
    smc.samples.ctx.timers(reset=True)
    t0 = time.perf_counter(); smc.step(); dt = time.perf_counter() - t0
    tm = smc.samples.ctx.timers()
    print(f"k={k} step={dt*1e3:.1f} ms nuts={tm[0]:.1f} ms leaps={smc.leapfrogs[k]} ({smc.leapfrogs[k]/N:.0f}/particle) -> {smc.leapfrogs[k]/tm[0]/1e6:.3f} G lf/s kernel; phi={smc.phi[k]:.4f} ess={smc.ess[k]:.1f}", flush=True)
