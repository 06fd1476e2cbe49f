"""Where the cold SMCSampler(K=50, N=65536, arma) construction and sample() go."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from smcnuts_amd import ArmaModel, SMCSampler, _capi
_capi.lib()
t = ArmaModel()
def tm(f, n=3):
    out = []
    for _ in range(n):
        t0 = time.perf_counter(); r = f(); out.append(time.perf_counter() - t0)
    return min(out) * 1e3, r
print("Context create+destroy    %.2f ms" % tm(lambda: _capi.Context(65536, t.model_id, t.model_data).close())[0])
print("np.full x_saved 107 MB    %.2f ms" % tm(lambda: np.full([51, 65536, 4], 0.0))[0])
print("np.empty x_saved          %.2f ms" % tm(lambda: np.empty([51, 65536, 4]))[0])
print("np.zeros x_saved          %.2f ms" % tm(lambda: np.zeros([51, 65536, 4]))[0])
for hist in (True, False):
    ms, s = tm(lambda: SMCSampler(K=50, N=65536, target=ArmaModel(), step_size=0.01, seed=11, save_history=hist), 2)
    print(f"SMCSampler(save_history={hist})  {ms:.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
s = SMCSampler(K=50, N=65536, target=ArmaModel(), step_size=0.01, seed=11)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
for trial in range(3):
    s = SMCSampler(K=50, N=65536, target=ArmaModel(), step_size=0.01, seed=12 + trial)
    t0 = time.perf_counter(); s.run_fused(); t1 = time.perf_counter(); s.finalise_async(download_history=False); t2 = time.perf_counter()
    s.download_history(); t3 = time.perf_counter()
    print(f"trial {trial}: run_fused {1e3*(t1-t0):.2f} ms, finalise (no history) {1e3*(t2-t1):.2f} ms, download_history {1e3*(t3-t2):.2f} ms; nuts kernels {s.samples.ctx.timers()[0]:.2f} ms in {int(s.samples.ctx.timers()[1])} launches, discarded {s.discarded_launches}")
