"""Why a process's second cold sample() is ~2 ms slower than its first: hold / free the first one's host arrays."""
import sys, time, gc
import numpy as np
sys.path.insert(0, ".")
from smcnuts_amd import ArmaModel, SMCSampler
keep = SMCSampler(K=25, N=65536, target=ArmaModel(), step_size=0.01, seed=1)
keep.run_fused(fuse_max=64); keep.finalise_async(download_history=False)
held = []
for mode in ("free", "free", "hold", "hold", "free", "hold"):
    t0 = time.perf_counter()
    s = SMCSampler(K=50, N=65536, target=ArmaModel(), step_size=0.01, seed=3)
    t1 = time.perf_counter()
    s.sample(show_progress=False)
    print(f"{mode}: construct {1e3*(t1-t0):.2f} ms run_time {1e3*s.run_time:.2f} ms  kernels {s.samples.ctx.timers()[0]:.2f} ms")
    s.samples.ctx.close()
    if mode == "hold":
        held.append((s.x_saved, s.logw_saved))
    del s
    gc.collect()
