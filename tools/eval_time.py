"""Time the target evaluation kernel alone (value + gradient for M particles)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smcnuts_amd import PRMwCDModel, ArmaModel
name = sys.argv[1] if len(sys.argv) > 1 else "prm"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
t = PRMwCDModel() if name == "prm" else ArmaModel()
rng = np.random.default_rng(0)
x = 0.3 * rng.standard_normal((M, t.dim))
t.logpdfgrad(x[:1024], 1.0)
for _ in range(3):
    t0 = time.perf_counter(); lp, g = t.logpdfgrad(x, 1.0); dt = time.perf_counter() - t0
    print(f"{name}: {M} evals incl. PCIe {dt*1e3:.1f} ms -> {M/dt/1e9:.3f} G eval/s (upper bound on time)", flush=True)
