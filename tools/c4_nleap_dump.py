import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smcnuts_amd import PRMwCDModel, SMCSampler
N, K = 65536, 5
smc = SMCSampler(K=K, N=N, target=PRMwCDModel(), step_size=0.01, lkernel="GaussianApproxLKernel", tempering=True, seed=5, save_history=False)
out = {}
for k in range(K):
    smc.step()
    st = smc.samples.ctx.tree_stats()
    out[f"nleap_{k}"] = st["nleap"].copy()
    out[f"idx_{k}"] = np.zeros(1)
    print(k, st["nleap"].mean(), st["nleap"].max(), smc.resampled[k] if hasattr(smc, "resampled") else None, flush=True)
np.savez_compressed(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "c4_nleap.npz"), **out)
