"""Interleaved A/B timing of the NUTS proposal in ONE process (cdna guide rule 24).

    python tools/ab_nuts.py [name=path/to/lib.so ...]      (default: the in-tree build only)
    env of a variant can be given as name=path:ENV=VAL:ENV2=VAL (applied while that lib launches)

Every variant gets its own context on the same steady-state arma particle set
(N = 65 536) and runs the same iteration key, so the work is identical; the time
is the HIP-event time of the NUTS kernel alone (smcn_timers)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smcnuts_amd import _capi, ArmaModel, SMCSampler, build

N = int(os.environ.get("AB_N", 65536))
REPS = int(os.environ.get("AB_REPS", 15))
variants = []
for a in sys.argv[1:]:
    name, rest = a.split("=", 1)
    parts = rest.split(":")
    env = dict(p.split("=", 1) for p in parts[1:])
    variants.append((name, parts[0], env))
if not variants:
    variants = [("tree", build.LIB, {})]

# steady-state particles from the default library
smc = SMCSampler(K=12, N=N, target=ArmaModel(), step_size=0.01, seed=10, save_history=False)
for _ in range(12):
    smc.step_async()
smc.samples.ctx.call("smcn_synchronize")
x = smc.samples.x
md = ArmaModel().model_data

ctxs = []
for name, path, env in variants:
    lib = C.CDLL(os.path.abspath(path))
    for fn, (args, res) in _capi.SIGNATURES.items():
        if hasattr(lib, fn):
            f = getattr(lib, fn); f.argtypes, f.restype = args, res
    h = C.c_void_p()
    assert lib.smcn_ctx_create(C.byref(h), 0, N, 0, _capi.MODEL_ARMA, _capi.dptr(md), md.size) == 0
    lib.smcn_set_seed(h, 10)
    xx = np.ascontiguousarray(x)
    lib.smcn_set_state(h, _capi.dptr(xx), _capi.dptr(np.zeros(N)))
    ctxs.append((name, lib, h, env))

def run(lib, h, env, it):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    t = np.zeros(6)
    lib.smcn_timers(h, _capi.dptr(t), 1)
    rc = lib.smcn_propose_nuts(h, 0.01, 1.0, 10, 100.0, it, None, None)
    assert rc == 0, lib.smcn_last_error(h)
    lib.smcn_timers(h, _capi.dptr(t), 1)
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    return t[0]

res = {n: [] for n, *_ in ctxs}
for rep in range(REPS + 2):
    for name, lib, h, env in ctxs:
        ms = run(lib, h, env, 100)
        if rep >= 2:
            res[name].append(ms)
leaps = None
for name, lib, h, env in ctxs:
    v = C.c_int64(0); lib.smcn_last_leapfrogs(h, C.byref(v)); leaps = v.value
    a = np.array(res[name])
    print(f"{name:14s} median {np.median(a)*1e3:8.1f} us   min {a.min()*1e3:8.1f} us   leapfrogs {leaps}   "
          f"{leaps/np.median(a)/1e6:.3f} G lf/s (kernel)")
