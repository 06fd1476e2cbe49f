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
