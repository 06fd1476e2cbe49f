"""Histogram of NUTS tree depths (doublings) and sizes on the headline workload, one launch per iteration.
    python tools/tree_depths.py [iterations]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smcnuts_amd import ArmaModel, SMCSampler

K = int(sys.argv[1]) if len(sys.argv) > 1 else 25
smc = SMCSampler(K=K, N=65536, target=ArmaModel(), step_size=0.01, seed=10, save_history=False)
hist = np.zeros(12, dtype=np.int64)
leaves = np.zeros(12, dtype=np.int64)
for k in range(K):
    smc.step()
    st = smc.samples.ctx.tree_stats()
    if k >= 5:
        d = np.clip(st["depth"], 0, 11)
        hist += np.bincount(d, minlength=12)
        leaves += np.bincount(d, weights=st["nleap"], minlength=12).astype(np.int64)
tot = hist.sum()
print("doublings  share of trees  share of leapfrogs  mean leapfrogs")
for d in range(12):
    if hist[d]:
        print(f"{d:9d}  {hist[d] / tot:14.5f}  {leaves[d] / leaves.sum():18.5f}  {leaves[d] / hist[d]:14.1f}")
