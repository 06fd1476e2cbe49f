import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smcnuts_amd import ArmaModel, SMCSampler
smc = SMCSampler(K=30, N=65536, target=ArmaModel(), step_size=0.01, seed=10, save_history=False)
prev = None
for k in range(16):
    smc.step()
    nl = smc.samples.ctx.tree_stats()["nleap"].astype(float)
    x = smc.samples.x
    if prev is not None and not smc.resampled[k]:
        c = np.corrcoef(prev, nl)[0, 1]
        # how well does "previous depth" rank the long trees?
        long_now = nl >= 31
        pred = prev >= 31
        print(k, "corr", round(c, 3), "P(long)", long_now.mean().round(3), "P(long|prev long)", long_now[pred].mean().round(3),
              "P(long|prev short)", long_now[~pred].mean().round(3))
    prev = nl
