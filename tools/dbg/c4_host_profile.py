"""Config 4's step-by-step loop: where does the HOST spend its share of an iteration?  cProfile over 20 iterations.
    python tools/dbg/c4_host_profile.py"""
import os, sys, cProfile, pstats, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smcnuts_amd import PRMwCDModel, SMCSampler
smc = SMCSampler(K=40, N=65536, target=PRMwCDModel(), step_size=0.01, lkernel="GaussianApproxLKernel", tempering=True, seed=10,
                 save_history=False)
for _ in range(14):
    smc.step()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(20):
    smc.step()
pr.disable()
t1 = time.perf_counter()
print(f"{(t1 - t0) / 20 * 1e3:.3f} ms per iteration under the profiler")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue()[:4000])
