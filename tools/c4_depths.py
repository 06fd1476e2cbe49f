"""Config 4: tree sizes per particle of consecutive iterations (for schedule studies):
   python tools/c4_depths.py [N] [K]  ->  gpurun_out/c4_nleap.npy  ([K][N] leapfrogs per particle)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ctypes as C
from smcnuts_amd import PRMwCDModel, SMCSampler
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
smc = SMCSampler(K=K, N=N, target=PRMwCDModel(), step_size=0.01, lkernel="GaussianApproxLKernel", tempering=True, seed=10,
                 save_history=False)
out = np.zeros((K, N), dtype=np.int32)
for k in range(K):
    smc.step()
    nl = np.zeros(N, dtype=np.int32)
    smc.samples.ctx.call("smcn_get_tree_stats", nl.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None)
    out[k] = nl
    h = np.bincount(np.floor(np.log2(nl + 1)).astype(int), minlength=12)
    print(k, "mean", nl.mean().round(1), "max", nl.max(), "depth histogram", h.tolist(), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/c4_nleap.npy", out)
