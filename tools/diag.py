"""GPU diagnostics (not a test): tree-size distribution and kernel time per iteration."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smcnuts_amd import ArmaModel, SMCSampler

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
smc = SMCSampler(K=30, N=N, target=ArmaModel(), step_size=0.01, seed=10, save_history=False)
for k in range(30):
    smc.samples.ctx.timers(reset=True)
    t0 = time.perf_counter()
    smc.step()
    dt = time.perf_counter() - t0
    tm = smc.samples.ctx.timers()
    st = smc.samples.ctx.tree_stats()
    nl = st["nleap"]
    if k in (0, 1, 2, 5, 10, 20, 29):
        h = np.bincount(st["depth"], minlength=12)
        print(f"k={k} step={dt*1e3:.3f}ms nuts={tm[0]:.3f}ms leaps={nl.sum()} mean={nl.mean():.2f} max={nl.max()} "
              f"p99={np.percentile(nl, 99):.0f} p999={np.percentile(nl, 99.9):.0f} depth_hist={h.tolist()} resampled={smc.resampled[k]}")
        # ideal schedule bound: groups=16384 concurrent, (nleap+1) evals each
        ev = nl.astype(np.int64) + 1
        print(f"     sum_evals/16384={ev.sum()/16384:.1f}  max_evals={ev.max()}  -> us/iter(lower bound)={tm[0]*1e3/max(ev.sum()/16384, ev.max()):.2f}")
