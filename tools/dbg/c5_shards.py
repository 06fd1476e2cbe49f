"""Repro: D = 256 over 8 in-process shards, twice; which generation / which rank differs?"""
import sys, threading
import numpy as np
sys.path.insert(0, ".")
from smcnuts_amd import IsoGaussian, SMCSampler
from smcnuts_amd.parallel import InProcessComm

def run(W, Nl, D, K, device=True, hook=None):
    g = InProcessComm(W)
    out = [None] * W
    def work(r):
        c = g.view(r)
        if not device:
            c.device_path = False
        s = SMCSampler(K=K, N=Nl * W, target=IsoGaussian(D), step_size=0.25, seed=77, save_history=False, comm=c)
        x0, lw0, _ = s.samples.ctx.get_state()
        s.sample(show_progress=False)
        out[r] = dict(ess=s.ess.copy(), x0nan=int(np.isnan(x0).sum()), lw0=(float(lw0.min()), float(lw0.max())),
                      x0abs=float(np.abs(x0).max()), x0zero=int((x0 == 0).sum()))
        s.samples.ctx.close()
    th = [threading.Thread(target=work, args=(r,)) for r in range(W)]
    [t.start() for t in th]; [t.join() for t in th]
    return out

for Nl in (4096, 131072):
    for trial in range(3):
        o = run(8, Nl, 256, 2)
        print(Nl, trial, "ess", o[0]["ess"], "same on ranks", all(np.array_equal(o[0]["ess"], q["ess"]) for q in o))
        for r, q in enumerate(o):
            if q["x0nan"] or q["x0zero"] or abs(q["lw0"][0]) > 1e-6 or abs(q["lw0"][1]) > 1e-6:
                print("   rank", r, q)
