import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smcnuts_amd import ArmaModel
from smcnuts_amd.proposal.nuts import NUTSProposal
out, eps = sys.argv[1], float(sys.argv[2])
N = 4096
rng = np.random.default_rng(1)
x = np.column_stack([0.1 * rng.standard_normal(N), 0.9 + 0.05 * rng.standard_normal(N), 0.2 * rng.standard_normal(N), -1.8 + 0.1 * rng.standard_normal(N)])
r = rng.standard_normal((N, 4))
prop = NUTSProposal(ArmaModel(), None, eps)
xn, rn = prop.rvs(x, r, 1.0, seed=7)
st = prop.last_stats
np.savez(out, xn=xn, rn=rn, nleap=st["nleap"], depth=st["depth"], ndraws=st["ndraws"])
print(out, "mean leaps", st["nleap"].mean(), "depth hist", np.bincount(st["depth"], minlength=12))
