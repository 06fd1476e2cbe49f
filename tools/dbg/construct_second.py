"""bench.py's end_to_end / end_to_end_second sequence with the library's set-up trace (-DSMCN_TRACE_SETUP build)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from smcnuts_amd import ArmaModel, SMCSampler, _capi
log = []
orig_call, orig_init = _capi.Context.call, _capi.Context.__init__
def call(self, name, *a):
    t = time.perf_counter(); r = orig_call(self, name, *a); d = 1e3 * (time.perf_counter() - t)
    if d > 0.2: log.append((name, round(d, 2)))
    return r
def init(self, *a, **k):
    t = time.perf_counter(); orig_init(self, *a, **k); log.append(("Context()", round(1e3 * (time.perf_counter() - t), 2)))
_capi.Context.call, _capi.Context.__init__ = call, init
keep = SMCSampler(K=25, N=65536, target=ArmaModel(), step_size=0.01, seed=1)
keep.run_fused(fuse_max=64); keep.finalise_async(download_history=False)
for trial in range(3):
    log.clear()
    t0 = time.perf_counter()
    s = SMCSampler(K=50, N=65536, target=ArmaModel(), step_size=0.01, seed=12 + trial)
    t1 = time.perf_counter()
    print(f"trial {trial}: construct {1e3*(t1-t0):.2f} ms", log, file=sys.stderr)
    s.sample(show_progress=False)
    t2 = time.perf_counter()
    s.samples.ctx.close()
    t3 = time.perf_counter()
    del s
    print(f"   run_time {1e3*s.run_time if False else 0:.2f} sample wall {1e3*(t2-t1):.2f} close {1e3*(t3-t2):.2f} del {1e3*(time.perf_counter()-t3):.2f}", file=sys.stderr)
