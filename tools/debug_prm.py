import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
from smcnuts_amd import PRMwCDModel
from smcnuts_amd.proposal.nuts import NUTSProposal
g = np.load("tests/golden/prmwcd_gaussL_temp.npz")
t = PRMwCDModel()
ot = orc.OracleTarget(orc.MODEL_PRMWCD, t.model_data, 13)
prop = NUTSProposal(t, None, float(g["eps"]))
for k in range(int(g["K"])):
    xn, rn = prop.rvs(g[f"x_in_{k}"], g[f"r_{k}"], float(g[f"phi_prop_{k}"]), tape=g[f"tape_{k}"], tape_off=g[f"tape_off_{k}"])
    st = prop.last_stats
    exp = np.diff(g[f"tape_off_{k}"])
    ref = orc.nuts_rvs(ot, g[f"x_in_{k}"], g[f"r_{k}"], float(g[f"phi_prop_{k}"]), float(g["eps"]), tape=g[f"tape_{k}"], tape_off=g[f"tape_off_{k}"])
    bad = np.flatnonzero(st["ndraws"] != exp)
    err = np.abs(xn - g[f"x_new_{k}"]).max(axis=1)
    print(k, "phi", float(g[f"phi_prop_{k}"]), "bad", bad.tolist(), "ndraws", st["ndraws"][bad].tolist(), exp[bad].tolist(),
          "nleap gpu/ref", st["nleap"][bad].tolist(), ref["nleap"][bad].tolist(), "depth", st["depth"][bad].tolist(), ref["depth"][bad].tolist(),
          "maxerr(ok)", np.delete(err, bad).max(), "max nleap", ref["nleap"].max(), "depth hist", np.bincount(ref["depth"], minlength=12).tolist())
    for b in bad:
        xi = g[f"x_in_{k}"][b]
        print("   x_in", xi, "lp gpu", t.logpdf(xi, float(g[f"phi_prop_{k}"])), "ref", ot.logpdf(xi, float(g[f"phi_prop_{k}"])))
