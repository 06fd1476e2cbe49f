"""A sampler constructed right after another one's close(): what waits for what (bench.py's end_to_end_second)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from smcnuts_amd import ArmaModel, SMCSampler, _capi
log = []
orig_call = _capi.Context.call
def call(self, name, *a):
    t = time.perf_counter(); r = orig_call(self, name, *a); d = 1e3 * (time.perf_counter() - t)
    if d > 0.2: log.append((name, round(d, 2)))
    return r
_capi.Context.call = call
keep = SMCSampler(K=25, N=65536, target=ArmaModel(), step_size=0.01, seed=1)
keep.run_fused(fuse_max=64); keep.finalise_async(download_history=False)
for gap_ms in (0, 1, 1, 0, 1, 21):
    s = SMCSampler(K=50, N=65536, target=ArmaModel(), step_size=0.01, seed=3)
    s.sample(show_progress=False)
    s.samples.ctx.close()
    if gap_ms % 2 == 0:
        hold = s                  # (the arrays are freed later: only the device side is gone)
    else:
        hold = None
    td = time.perf_counter()
    del s
    td = 1e3 * (time.perf_counter() - td)
    time.sleep((gap_ms // 2 * 2) / 1e3)
    log.clear()
    t0 = time.perf_counter()
    s2 = SMCSampler(K=50, N=65536, target=ArmaModel(), step_size=0.01, seed=4)
    print(f"gap {gap_ms // 2 * 2:3d} ms after close(), arrays {'held' if hold is not None else f'freed ({td:.1f} ms)'}: construct {1e3*(time.perf_counter()-t0):.2f} ms", log)
    s2.samples.ctx.close()
    del s2, hold
    time.sleep(0.1)
