import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smcnuts_amd import ArmaModel
from smcnuts_amd.proposal.nuts import NUTSProposal
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "arma_fwd.npz"))
prop = NUTSProposal(ArmaModel(), None, float(g["eps"]))
for k in range(int(g["K"])):
    xn, rn = prop.rvs(g[f"x_in_{k}"], g[f"r_{k}"], float(g[f"phi_prop_{k}"]), tape=g[f"tape_{k}"], tape_off=g[f"tape_off_{k}"])
    st = prop.last_stats
    want = np.diff(g[f"tape_off_{k}"])
    bad = np.nonzero(st["ndraws"] != want)[0]
    deep = np.nonzero(st["depth"] >= 5)[0]
    print(k, "bad", bad.tolist(), "got ndraws", st["ndraws"][bad].tolist(), "want", want[bad].tolist(), "depth got", st["depth"][bad].tolist(),
          "nleap got", st["nleap"][bad].tolist(), "| #depth>=5:", len(deep), "max depth", st["depth"].max(), flush=True)
