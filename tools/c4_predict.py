"""Config 4: is the tree size of a particle predictable from what is known BEFORE the NUTS launch (x, r)?
   python tools/c4_predict.py  ->  gpurun_out/c4_pred.npz (x, r, nleap of one steady-state iteration)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ctypes as C
from smcnuts_amd import PRMwCDModel, SMCSampler
N, K = 65536, 14
smc = SMCSampler(K=K, N=N, target=PRMwCDModel(), step_size=0.01, lkernel="GaussianApproxLKernel", tempering=True, seed=10,
                 save_history=False)
for k in range(K - 1):
    smc.step()
s = smc.samples
s.normalise_weights(); s.calculate_ess(); s.resample_if_required()
x = s.x.copy()
s.propose_samples()
r, x_new, r_new, _ = s.ctx.get_proposal()
nl = np.zeros(N, dtype=np.int32)
s.ctx.call("smcn_get_tree_stats", nl.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None)
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/c4_pred.npz", x=x, r=r, nleap=nl, phi=s.phi_new)
print("mean", nl.mean(), "max", nl.max(), "phi", s.phi_new)
