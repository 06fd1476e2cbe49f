import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
from smcnuts_amd import ArmaModel
from smcnuts_amd.proposal.nuts import NUTSProposal
md = int(sys.argv[1]); N = int(sys.argv[2]); eps = float(sys.argv[3])
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "smcnuts_amd", "model", "data")
ot = orc.OracleTarget(orc.MODEL_ARMA, orc.arma_data(os.path.join(DATA, "arma.json")), 4)
rng = np.random.default_rng(1)
x = np.column_stack([0.1 * rng.standard_normal(N), 0.9 + 0.05 * rng.standard_normal(N), 0.2 * rng.standard_normal(N), -1.8 + 0.1 * rng.standard_normal(N)])
r = rng.standard_normal((N, 4))
prop = NUTSProposal(ArmaModel(), None, eps, max_depth=md)
print("launch", flush=True)
xn, rn = prop.rvs(x, r, 1.0, seed=7)
st = prop.last_stats
print("done; depth hist", np.bincount(st["depth"], minlength=12), flush=True)
ref = orc.nuts_rvs(ot, x, r, 1.0, eps, seed=7, iteration=0, max_depth=md)
bad = np.nonzero(st["nleap"] != ref["nleap"])[0]
print("mismatching nleap:", len(bad), bad[:20].tolist(), "max|dx|", np.abs(xn - ref["x_new"]).max())
print("per 64-particle chunk:", [int(((bad >= a) & (bad < a + 64)).sum()) for a in range(0, N, 64)][:32])
print("got/ref depth of bad:", list(zip(st["depth"][bad][:12].tolist(), ref["depth"][bad][:12].tolist())), "nleap", list(zip(st["nleap"][bad][:12].tolist(), ref["nleap"][bad][:12].tolist())))

print("lpri0 got/ref:", [(round(float(a),6), round(float(b),6)) for a, b in zip(st["lpri0"][bad][:6], ref["lpri0"][bad][:6])])
print("llik0 got/ref:", [(round(float(a),6), round(float(b),6)) for a, b in zip(st["llik0"][bad][:6], ref["llik0"][bad][:6])])
print("ndraws got/ref:", list(zip(st["ndraws"][bad][:12].tolist(), ref["ndraws"][bad][:12].tolist())))
# does a bad particle's result equal another particle's reference (index confusion)?
for b_ in bad[:12]:
    hit = np.nonzero((ref["nleap"] == st["nleap"][b_]) & (np.abs(ref["x_new"] - xn[b_]).max(axis=1) < 1e-9))[0]
    print(int(b_), "matches reference of particle(s)", hit.tolist())

for b_ in bad[:12]:
    hit = np.nonzero((np.abs(ref["llik0"] - st["llik0"][b_]) < 1e-6 * (1 + np.abs(ref["llik0"]))) & (np.abs(ref["lpri0"] - st["lpri0"][b_]) < 1e-6))[0]
    print(int(b_), "start density equals that of particle(s)", hit.tolist())

first = bad - 64
print("first-unit (p-64) ref depth:", ref["depth"][first].tolist(), "moved:", (np.abs(ref["x_new"][first] - x[first]).max(axis=1) > 0).tolist())
print("depth-1 first units overall:", np.nonzero(ref["depth"][:64] == 1)[0].tolist())
lp_new = ot.parts(ref["x_new"][first]) if hasattr(ot, "parts") else None
if lp_new is not None:
    print("got lpri0 vs density at x_new(p-64):", [(round(float(a), 6), round(float(b), 6)) for a, b in zip(st["lpri0"][bad], lp_new[0])])

if os.environ.get("SMCN_LIB", "").endswith("lib_dbg.so"):
    good = np.setdiff1d(np.arange(N), bad)[:4]
    for b_ in good:
        print("good", int(b_), "dbg x[0:2] at INIT", rn[b_][:2], "p", rn[b_][2], "slot[EM]", rn[b_][3], "| x0[p]", x[b_][:2])
    print("NaN records:", np.nonzero(np.isnan(rn).any(axis=1))[0].tolist())
    for b_ in bad[:12]:
        print(int(b_), "dbg x[0:2] at INIT", rn[b_][:2], "p", rn[b_][2], "slot[EM]", rn[b_][3], "| x0[p]", x[b_][:2], "x0[p-64]", x[b_ - 64][:2])

print("bad got (depth,nleap,ndraws):", list(zip(st["depth"][bad].tolist(), st["nleap"][bad].tolist(), st["ndraws"][bad].tolist())))
print("ref of p-64 (depth,nleap,ndraws):", list(zip(ref["depth"][bad-64].tolist(), ref["nleap"][bad-64].tolist(), ref["ndraws"][bad-64].tolist())))
print("got x_new[bad][:2] vs x0[p]:", [(np.round(xn[b_][:2], 5).tolist(), np.round(x[b_][:2], 5).tolist()) for b_ in bad[:4]])
print("got llik1 vs ref llik1 of p:", [(round(float(a), 4), round(float(b), 4)) for a, b in zip(st["llik1"][bad][:6], ref["llik1"][bad][:6])])
