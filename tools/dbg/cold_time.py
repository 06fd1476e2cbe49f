import sys, time
import numpy as np
sys.path.insert(0, ".")
from smcnuts_amd import ArmaModel, SMCSampler
import smcnuts_amd.smc_sampler as S
for trial in range(3):
    t0 = time.perf_counter()
    s = SMCSampler(K=50, N=65536, target=ArmaModel(), step_size=0.01, seed=12 + trial)
    t1 = time.perf_counter()
    log = []
    orig_dl, orig_ready = s._download_validated, s._history_ready
    def dl(k, orig=orig_dl):
        a = time.perf_counter(); up = s._dl_upto; orig(k); log.append(("dl", up + 1, k, 1e3 * (time.perf_counter() - a)))
    def ready(orig=orig_ready):
        a = time.perf_counter(); orig(); d = 1e3 * (time.perf_counter() - a)
        if d > 0.05: log.append(("join", d))
    s._download_validated, s._history_ready = dl, ready
    orig_call = s.samples.ctx.call
    def call(name, *a):
        t = time.perf_counter(); r = orig_call(name, *a); d = 1e3 * (time.perf_counter() - t)
        if d > 0.3: log.append((name, round(d, 2)))
        return r
    s.samples.ctx.call = call
    s.sample(show_progress=False)
    t2 = time.perf_counter()
    print(f"trial {trial}: construct {1e3*(t1-t0):.2f} ms, sample {1e3*(t2-t1):.2f} ms, run_time {1e3*s.run_time:.2f}; nuts {s.samples.ctx.timers()[0]:.2f} ms / {int(s.samples.ctx.timers()[1])} launches")
    for e in log: print("   ", e)
