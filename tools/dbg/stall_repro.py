"""bench.py's sequence around the cold sample() (a main sampler alive with its history; cold samplers built, run, closed,
freed one after another), repeated: which constructor stalls, and -- under rocprofv3 --hip-trace --hsa-trace -- on what.
`probe` = a tiny kernel + an 8-byte device-to-host copy on the MAIN sampler's stream (smcn_last_leapfrogs): if the probe
is slow after a teardown step, the GPU queues of the whole process were held up by that step, not one sampler's stream."""
import ctypes as C, gc, sys, time
sys.path.insert(0, ".")
import numpy as np
from smcnuts_amd import ArmaModel, SMCSampler

main = SMCSampler(K=25, N=65536, target=ArmaModel(), step_size=0.01, seed=10)
main.run_fused(upto=25, fuse_max=64)
main.finalise_async(download_history=False)


def probe():
    t = time.perf_counter()
    main.samples.ctx.last_leapfrogs()
    return 1e3 * (time.perf_counter() - t)


probe()
mode = sys.argv[2] if len(sys.argv) > 2 else "plain"
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    p0 = probe()
    t0 = time.perf_counter()
    cold = SMCSampler(K=50, N=65536, target=ArmaModel(), step_size=0.01, seed=11 + trial)
    t1 = time.perf_counter()
    cold.sample(show_progress=False)
    t2 = time.perf_counter()
    p1 = probe()
    cold.samples.ctx.close()
    p2 = probe()
    if mode == "keep":
        keep = getattr(sys.modules[__name__], "_keep", [])
        keep.append(cold.x_saved)            # the host history is NOT given back (no munmap)
        sys.modules[__name__]._keep = keep
    del cold
    gc.collect()
    p3 = probe()
    time.sleep(0.02)
    p4 = probe()
    print(f"trial {trial}: construct {1e3 * (t1 - t0):7.2f} ms  sample {1e3 * (t2 - t1):6.2f} ms | probe before {p0:6.2f}, after sample {p1:6.2f}, "
          f"after close {p2:6.2f}, after del+gc {p3:6.2f}, 20 ms later {p4:6.2f}", flush=True)
