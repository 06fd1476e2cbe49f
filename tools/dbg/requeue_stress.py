"""The inner park level of config 4's first NUTS launch, checked on whole runs: every output of a tempered PRMwCD run with the
level on (nuts_cap = (9, True, 8), shipped) equals the run with it off, bit for bit.  ~1.2 M hand-overs.
    python tools/dbg/requeue_stress.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from smcnuts_amd import PRMwCDModel, SMCSampler


def run(N, K, requeue, seed):
    smc = SMCSampler(K=K, N=N, target=PRMwCDModel(), step_size=0.01, lkernel="GaussianApproxLKernel", tempering=True, seed=seed,
                     nuts_cap=(9, True, requeue), save_history=False)
    for _ in range(K):
        smc.step()
    smc.finalise()
    x, logw, _ = smc.samples.ctx.get_state()
    return smc, x, logw


for N, K, seed in ((65536, 14, 10), (8192, 20, 3), (1000, 30, 5)):
    a, xa, wa = run(N, K, 0, seed)
    b, xb, wb = run(N, K, 8, seed)
    same = np.array_equal(xa, xb) and np.array_equal(wa, wb)
    for name in ("ess", "phi", "mean_estimate", "variance_estimate", "leapfrogs", "log_likelihood"):
        same = same and np.array_equal(getattr(a, name), getattr(b, name))
    print(f"N {N} K {K}: leapfrogs {int(np.sum(a.leapfrogs))}, identical {same}")
    assert same
