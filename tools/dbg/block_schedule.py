"""The block schedule of a cold SMCSampler(K=50, N=65536, arma).sample(): (k0, B) per NUTS launch, what each cost."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from smcnuts_amd import ArmaModel, SMCSampler, _capi
log = []
orig_call = _capi.Context.call
def call(self, name, *a):
    t = time.perf_counter(); r = orig_call(self, name, *a); d = 1e3 * (time.perf_counter() - t)
    if name in ("smcn_block_launch", "smcn_block_wait", "smcn_history_download", "smcn_fuse_decide", "smcn_block_resample_local") or d > 0.3:
        log.append((name, [x for x in a[:2] if isinstance(x, int)], round(d, 2)))
    return r
_capi.Context.call = call
keep = SMCSampler(K=25, N=65536, target=ArmaModel(), step_size=0.01, seed=1)
keep.run_fused(fuse_max=64); keep.finalise_async(download_history=False)
for trial in range(3):
    s = SMCSampler(K=50, N=65536, target=ArmaModel(), step_size=0.01, seed=12 + trial)
    log.clear()
    t0 = time.perf_counter()
    s.sample(show_progress=False)
    print(f"trial {trial}: run_time {1e3*s.run_time:.2f} ms (wall {1e3*(time.perf_counter()-t0):.2f}); resampled at {[k for k, r in enumerate(s.resampled) if r]}; "
          f"nuts kernels {s.samples.ctx.timers()[0]:.2f} ms in {int(s.samples.ctx.timers()[1])} launches")
    for e in log: print("   ", e)
