import sys, time, runpy
sys.path.insert(0, "/root/repo")
from smcnuts_amd import _capi
orig = _capi.Context.call
T0 = time.perf_counter()
def call(self, name, *a):
    t = time.perf_counter(); r = orig(self, name, *a); d = 1e3 * (time.perf_counter() - t)
    if d > 0.5 or name in ("smcn_get_state", "smcn_ctx_destroy", "smcn_fast_begin"):
        print(f"[{1e3*(t-T0):9.1f} ms] ctx {id(self) % 10000:4d} {name} {d:.2f} ms", file=sys.stderr)
    return r
_capi.Context.call = call
_gs = _capi.Context.get_state
def get_state(self, *a, **k):
    import numpy as np
    t = time.perf_counter(); orig(self, "smcn_synchronize"); d1 = 1e3 * (time.perf_counter() - t)
    t = time.perf_counter(); lw = np.empty(self.N); orig(self, "smcn_get_state", None, _capi.dptr(lw), None); d2 = 1e3 * (time.perf_counter() - t)
    t = time.perf_counter(); r = _gs(self, *a, **k); d3 = 1e3 * (time.perf_counter() - t)
    print(f"[{1e3*(t-T0):9.1f} ms] ctx {id(self) % 10000:4d} get_state: synchronize {d1:.2f} ms, logw only (0.5 MB) {d2:.2f} ms, full {d3:.2f} ms", file=sys.stderr)
    return r
_capi.Context.get_state = get_state
import threading
_run = threading.Thread.run
def run(self):
    t = time.perf_counter()
    print(f"[{1e3*(t-T0):9.1f} ms] thread {self.name} starts", file=sys.stderr)
    _run(self)
    print(f"[{1e3*(time.perf_counter()-T0):9.1f} ms] thread {self.name} ends after {1e3*(time.perf_counter()-t):.1f} ms", file=sys.stderr)
threading.Thread.run = run
sys.argv = ["bench.py"] + (sys.argv[1:] or ["--steps", "20", "--warmup", "5"]) + ["--no-cpu-baseline"]
runpy.run_path("/root/repo/bench.py", run_name="__main__")
