"""oracle/cpu_baseline.py -- TEST / BENCH INFRASTRUCTURE ONLY.  NOT PRODUCT CODE.

The CPU timings bench.py reports beside the GPU number: the NUTS proposal on
a particle state handed over by the bench (the GPU run's own post-warm-up particles), for a
bounded time budget each:

  c1   oracle/smcnuts_oracle.c, one thread (the reference is single-threaded)
  call the same on every core this process may use (particles split over worker processes)
  py   oracle/pynuts.py: the reference-shaped serial Python loop, density through ctypes

    python oracle/cpu_baseline.py <state.npy> <model_data.npy> <seed> <budget_s>   -> one JSON line
"""
import json
import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != _HERE]   # (run as a script: `oracle` must be the package)
sys.path.insert(0, os.path.dirname(_HERE))
from oracle import oracle as orc          # noqa: E402
from oracle.pynuts import PyNUTS          # noqa: E402


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _worker(args):
    x, md, seed, it, base = args
    ot = orc.OracleTarget(orc.MODEL_ARMA, md, 4)
    r = orc.philox_normals(seed, 1000 + it, x.shape[0], 4, 1, particle_base=base)
    t = time.perf_counter()
    res = orc.nuts_rvs(ot, x, r, 1.0, 0.01, seed=seed, iteration=2000 + it, particle_base=base)
    return int(res["nleap"].sum()), time.perf_counter() - t


def main():
    x = np.load(sys.argv[1])
    md = np.load(sys.argv[2])
    seed, budget = int(sys.argv[3]), float(sys.argv[4])
    ot = orc.OracleTarget(orc.MODEL_ARMA, md, 4)
    ncpu = len(os.sched_getaffinity(0))
    out = {"cpu_model": cpu_model(), "nproc": os.cpu_count(), "usable_cores": ncpu}

    # c1: the C port on one thread
    sub = min(x.shape[0], 16384)
    moms = [orc.philox_normals(seed, 1000 + k, sub, 4, 1) for k in range(2)]
    leaps, t_n, reps = 0, 0.0, 0
    while t_n < budget:
        t = time.perf_counter()
        res = orc.nuts_rvs(ot, x[:sub], moms[reps % 2], 1.0, 0.01, seed=seed, iteration=2000 + reps)
        t_n += time.perf_counter() - t
        leaps += int(res["nleap"].sum())
        reps += 1
    out["c_one_core"] = {"value": leaps / t_n, "unit": "leapfrog/s", "cores": 1,
                         "sample": f"{reps} x {sub} particles, {leaps} leapfrogs, {t_n:.1f} s"}

    # call: the C port on all usable cores, particles split over worker processes (wall clock)
    import multiprocessing as mp
    W = max(1, min(ncpu, 64))
    per = max(256, min(x.shape[0] // W, 8192))
    leaps, wall, rounds = 0, 0.0, 0
    with mp.get_context("fork").Pool(W) as pool:
        pool.map(_worker, [(x[:64], md, seed, 0, 0)] * W)          # start-up (library load) outside the clock
        while wall < budget:
            jobs = [(x[(w * per) % (x.shape[0] - per + 1):(w * per) % (x.shape[0] - per + 1) + per], md, seed, rounds, w * per)
                    for w in range(W)]
            t = time.perf_counter()
            res = pool.map(_worker, jobs)
            wall += time.perf_counter() - t
            leaps += sum(a for a, _ in res)
            rounds += 1
    out["c_all_cores"] = {"value": leaps / wall, "unit": "leapfrog/s", "cores": W,
                          "sample": f"{rounds} rounds x {W} workers x {per} particles, {leaps} leapfrogs, {wall:.1f} s wall"}

    # py: the reference-shaped Python loop, one core
    prop = PyNUTS(ot, 0.01, rng=np.random.RandomState(seed))
    n_py, t_p, chunk = 0, 0.0, 32
    rr = orc.philox_normals(seed, 3000, min(x.shape[0], 4096), 4, 1)
    while t_p < budget and n_py + chunk <= rr.shape[0]:
        t = time.perf_counter()
        prop.rvs(x[n_py:n_py + chunk], rr[n_py:n_py + chunk], 1.0)
        t_p += time.perf_counter() - t
        n_py += chunk
    out["python_serial"] = {"value": prop.nleap / t_p, "unit": "leapfrog/s", "cores": 1,
                            "sample": f"{n_py} particles, {prop.nleap} leapfrogs, {t_p:.1f} s; tree logic in Python as "
                                      "smcnuts/proposal/nuts.py, density through ctypes (2 calls per leapfrog)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
