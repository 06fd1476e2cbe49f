import os, sys
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle as orc
from smcnuts_amd import PRMwCDModel, _capi
DATA = "smcnuts_amd/model/data"
g = np.load("tests/golden/prmwcd_gaussL_temp.npz")
t = PRMwCDModel(); ot = orc.OracleTarget(orc.MODEL_PRMWCD, orc.prmwcd_data(os.path.join(DATA, "PRMwCD.json")), 13)
x = np.concatenate([g["x_saved"][k] for k in range(int(g["K"]) + 1)])[:192]
N = x.shape[0]
ctx = _capi.Context(N, t.model_id, t.model_data); ctx.set_seed(404); ctx.set_state(x=x, logw=np.zeros(N))
for phi, it, eps in ((1.0, 0, 3e-4), (0.2, 1, 1e-3), (0.2, 2, 2e-4), (1.0, 3, 1e-4)):
    ctx.propose_nuts(eps, phi, it)
    r, xn, rn, _ = ctx.get_proposal(); st = ctx.tree_stats()
    ref = orc.nuts_rvs(ot, x, r, phi, eps, seed=404, iteration=it)
    e = np.abs(xn - ref["x_new"]).max(axis=1)
    print(phi, eps, "ndraws mismatches", int((st["ndraws"] != ref["ndraws"]).sum()), "nleap>=1023", int((st["nleap"] >= 1023).sum()),
          "depth<=9", int((st["depth"] <= 9).sum()), "err quantiles", np.quantile(e, [0.5, 0.9, 0.99, 1.0]), "n>1e-9", int((e > 1e-9).sum()))
