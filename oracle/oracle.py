"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.  NOT PRODUCT CODE.

CPU restatement of the SMC-NUTS hot path of the reference, used only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker
(never imported by smcnuts_amd/).

* NUTS transition + target densities: plain C in oracle/smcnuts_oracle.c
  (loaded here through ctypes).
* Weight path (normalise, ESS, multinomial resampling, L-kernels, ESS
  tempering, estimates) and the SMC loop order: NumPy, below, each function
  citing the reference file:line it follows.

Parity status: the NUTS transition, the weight path and the loop order are
pinned against golden vectors produced by the real reference
(tests/golden/make_golden.py; checked in tests/test_oracle_golden.py).  The
arma / PRMwCD densities are PARITY UNPINNED (BridgeStan is absent, see the
header of smcnuts_oracle.c).
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
SRC_PATH = os.path.join(HERE, "smcnuts_oracle.c")

MODEL_GAUSS, MODEL_ARMA, MODEL_PRMWCD = 0, 1, 2

_lib = None


def build(force=False):
    """gcc the C restatement into oracle/liboracle.so (idempotent)."""
    if (not force and os.path.exists(LIB_PATH)
            and os.path.getmtime(LIB_PATH) >= os.path.getmtime(SRC_PATH)):
        return LIB_PATH
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", LIB_PATH, SRC_PATH, "-lm"]
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) or (
                os.path.exists(SRC_PATH) and os.path.getmtime(LIB_PATH) < os.path.getmtime(SRC_PATH)):
            build()
        _lib = C.CDLL(LIB_PATH)
        dp, ip, lp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        _lib.oracle_target_eval.argtypes = [C.c_int, dp, C.c_int64, C.c_int, dp, C.c_double, dp, dp, dp, dp]
        _lib.oracle_constrain.argtypes = [C.c_int, C.c_int64, C.c_int, dp, dp]
        _lib.oracle_nuts_rvs.argtypes = [
            C.c_int, dp, C.c_int64, C.c_int, dp, dp, C.c_double, C.c_double, C.c_int, C.c_double,
            C.c_int, dp, lp, C.c_uint64, C.c_uint32, C.c_int64, dp, dp, dp, dp, dp, dp, ip, ip, ip, ip]
        _lib.oracle_philox_uniforms.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                                C.c_uint32, C.c_int64, dp]
        _lib.oracle_philox_normals.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, dp]
        _lib.oracle_philox_particle_uniforms.argtypes = [C.c_uint64, C.c_uint32, C.c_int64, C.c_int64,
                                                         C.c_uint32, C.c_uint32, dp]
        _lib.oracle_philox_raw.argtypes = [C.POINTER(C.c_uint32)] * 3
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


# ---------------------------------------------------------------------------
# model data (flat double arrays; layouts documented in smcnuts_oracle.c)
# ---------------------------------------------------------------------------
def gauss_data(D, prior_sd=1.0, lik_mean=None, lik_sd=1.0):
    has = 0.0 if lik_mean is None else 1.0
    return np.array([D, prior_sd, has, 0.0 if lik_mean is None else lik_mean, lik_sd], dtype=np.float64)


def arma_data(path):
    d = json.load(open(path))
    return np.concatenate([[float(d["T"])], np.asarray(d["y"], dtype=np.float64)])


def prmwcd_data(path):
    d = json.load(open(path))
    return np.concatenate([[float(d["N"]), float(d["M"]), float(d["Clength"]), float(d["q"])],
                           np.asarray(d["y"], dtype=np.float64),
                           np.asarray(d["Xkernel"], dtype=np.float64)])


class OracleTarget:
    """Duck-typed target with the StanModel surface (model/bridgestan.py:28-120)
    backed by the C restatement.  Used to drive the REAL reference when making
    golden vectors and as the checker's target."""

    def __init__(self, model, data, D):
        self.model, self.data, self.dim = model, np.ascontiguousarray(data, dtype=np.float64), int(D)
        self.constrained_dim = int(D)

    def parts(self, x):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        M = x.shape[0]
        lpri, llik = np.empty(M), np.empty(M)
        lib().oracle_target_eval(self.model, _dp(self.data), M, self.dim, _dp(x), 1.0, None, None,
                                 _dp(lpri), _dp(llik))
        return lpri, llik

    def logpdf(self, x, phi=1.0):
        x = np.asarray(x, dtype=np.float64)
        x2 = np.ascontiguousarray(np.atleast_2d(x))
        out = np.empty(x2.shape[0])
        lib().oracle_target_eval(self.model, _dp(self.data), x2.shape[0], self.dim, _dp(x2), float(phi),
                                 _dp(out), None, None, None)
        return float(out[0]) if x.ndim == 1 else out

    def logpdfgrad(self, x, phi=1.0):
        x = np.asarray(x, dtype=np.float64)
        x2 = np.ascontiguousarray(np.atleast_2d(x))
        g = np.empty_like(x2)
        lib().oracle_target_eval(self.model, _dp(self.data), x2.shape[0], self.dim, _dp(x2), float(phi),
                                 None, _dp(g), None, None)
        return g[0] if x.ndim == 1 else g

    def constrain(self, x):
        x = np.asarray(x, dtype=np.float64)
        x2 = np.ascontiguousarray(np.atleast_2d(x))
        out = np.empty_like(x2)
        lib().oracle_constrain(self.model, x2.shape[0], self.dim, _dp(x2), _dp(out))
        return out[0] if x.ndim == 1 else out


# ---------------------------------------------------------------------------
# NUTS (C)
# ---------------------------------------------------------------------------
def load_variant(path):
    """Another build of smcnuts_oracle.c (tests: other compiler flags) with the NUTS entry point bound."""
    v = C.CDLL(path)
    v.oracle_nuts_rvs.argtypes = lib().oracle_nuts_rvs.argtypes
    return v


def nuts_rvs(target, x, r, phi, eps, max_depth=10, delta_max=100.0, tape=None, tape_off=None,
             seed=0, iteration=0, particle_base=0, clib=None):
    """NUTSProposal.rvs (proposal/nuts.py:34-56).  tape mode if `tape` given,
    else Philox(seed, iteration, particle)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    r = np.ascontiguousarray(r, dtype=np.float64)
    N, D = x.shape
    xn, rn = np.empty_like(x), np.empty_like(r)
    lp0, ll0, lp1, ll1 = (np.empty(N) for _ in range(4))
    nleap, depth, ndraws, flags = (np.empty(N, dtype=np.int32) for _ in range(4))
    ip = C.POINTER(C.c_int32)
    if tape is not None:
        tape = np.ascontiguousarray(tape, dtype=np.float64)
        tape_off = np.ascontiguousarray(tape_off, dtype=np.int64)
        mode, tp, to = 0, _dp(tape), tape_off.ctypes.data_as(C.POINTER(C.c_int64))
    else:
        mode, tp, to = 1, None, None
    rc = (clib or lib()).oracle_nuts_rvs(target.model, _dp(target.data), N, D, _dp(x), _dp(r), float(phi),
                               float(eps), int(max_depth), float(delta_max), mode, tp, to,
                               int(seed), int(iteration), int(particle_base), _dp(xn), _dp(rn),
                               _dp(lp0), _dp(ll0), _dp(lp1), _dp(ll1),
                               nleap.ctypes.data_as(ip), depth.ctypes.data_as(ip),
                               ndraws.ctypes.data_as(ip), flags.ctypes.data_as(ip))
    if rc != 0:
        raise RuntimeError("oracle_nuts_rvs failed")
    return dict(x_new=xn, r_new=rn, lpri0=lp0, llik0=ll0, lpri1=lp1, llik1=ll1, nleap=nleap,
                depth=depth, ndraws=ndraws, flags=flags)


def philox_uniforms(seed, iteration, particle, stream, q0, count):
    out = np.empty(count)
    lib().oracle_philox_uniforms(int(seed), int(iteration), int(particle), int(stream), int(q0), count, _dp(out))
    return out


def philox_particle_uniforms(seed, iteration, particle_base, N, stream, q):
    """Draw q of every particle's `stream` (one uniform per particle)."""
    out = np.empty(N)
    lib().oracle_philox_particle_uniforms(int(seed), int(iteration), int(particle_base), N, int(stream),
                                          int(q), _dp(out))
    return out


def philox_normals(seed, iteration, N, D, stream, particle_base=0):
    out = np.empty((N, D))
    row = np.empty(D)
    for i in range(N):
        lib().oracle_philox_normals(int(seed), int(iteration), int(particle_base + i), int(stream), D, _dp(row))
        out[i] = row
    return out


def philox_raw(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().oracle_philox_raw(c, k, o)
    return list(o)


# ---------------------------------------------------------------------------
# Weight path (NumPy)
# ---------------------------------------------------------------------------
def logsumexp_scipy(a):
    """scipy.special.logsumexp as of SciPy 1.15.3 (`_logsumexp`): the maximum
    elements are taken out of the sum; out = log1p(s/m) + log(m) + max."""
    a = np.asarray(a, dtype=np.float64)
    if a.size == 0:
        return -np.inf
    amax = np.max(a)
    is_max = a == amax
    m = float(np.sum(is_max))
    shift = amax if np.isfinite(amax) else 0.0
    with np.errstate(all="ignore"):
        e = np.exp(np.where(is_max, -np.inf, a) - shift)
        s = np.sum(e)
        s = s if s == 0 else s / m
        return float(np.log1p(s) + np.log(m) + amax)


def normalise_weights(logw):
    """Samples.normalise_weights (samples/samples.py:91-105)."""
    index = ~np.isneginf(logw)
    ll = logsumexp_scipy(logw[index])
    wn = np.zeros_like(logw)
    with np.errstate(all="ignore"):
        wn[index] = np.exp(logw[index] - ll)
    return wn, ll


def calculate_ess(wn):
    """Samples.calculate_ess (samples/samples.py:108-113)."""
    return 1.0 / np.sum(np.square(wn))


SCAN_TILE, SCAN_PER_THREAD, WAVE = 1024, 4, 64


def blocked_cumsum(w):
    """Inclusive prefix sum in the fixed blocked order the HIP scan kernel uses
    (DESIGN.md "resample"): 1024-element tiles; 4 consecutive elements per
    thread summed sequentially; Hillis-Steele across the 64 lanes of a wave;
    the 4 wave totals of a tile and the tile totals combined sequentially."""
    w = np.asarray(w, dtype=np.float64)
    N = w.size
    nt = (N + SCAN_TILE - 1) // SCAN_TILE
    a = np.zeros(nt * SCAN_TILE)
    a[:N] = w
    a = a.reshape(nt, SCAN_TILE // (WAVE * SCAN_PER_THREAD), WAVE, SCAN_PER_THREAD)
    s = np.cumsum(a, axis=3)                  # sequential per thread
    v = s[..., -1].copy()                     # lane totals [nt, 4, 64]
    k = 1
    while k < WAVE:
        nv = v.copy()
        nv[..., k:] = v[..., k:] + v[..., :-k]
        v = nv
        k *= 2
    excl = np.zeros_like(v)
    excl[..., 1:] = v[..., :-1]
    wtot = v[..., -1]                         # [nt, 4]
    woff = np.zeros_like(wtot)
    for j in range(1, wtot.shape[1]):
        woff[:, j] = woff[:, j - 1] + wtot[:, j - 1]
    local = (woff[..., None] + excl)[..., None] + s          # [nt,4,64,4]
    local = local.reshape(nt, SCAN_TILE)
    ttot = local[:, -1]
    toff = np.zeros(nt)
    for b in range(1, nt):
        toff[b] = toff[b - 1] + ttot[b - 1]
    return (toff[:, None] + local).reshape(-1)[:N]


def _searchsorted_right(cdf, u):
    """Plain bisection, identical to the device search."""
    out = np.empty(u.size, dtype=np.int64)
    n = cdf.size
    for i, key in enumerate(u):
        lo, hi = 0, n
        while lo < hi:
            mid = lo + ((hi - lo) >> 1)
            if key < cdf[mid]:
                hi = mid
            else:
                lo = mid + 1
        out[i] = lo
    return out


def multinomial_indices(wn, u, order="sequential"):
    """rng.choice(arange(N), N, p=wn) (samples/samples.py:138-139) given its N
    uniforms: cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, 'right')."""
    if order == "sequential":
        cdf = np.cumsum(wn)
        cdf = cdf / cdf[-1]
        return np.searchsorted(cdf, u, side="right")
    cdf = blocked_cumsum(wn)
    cdf = cdf / cdf[-1]
    return _searchsorted_right(cdf, np.asarray(u))


def systematic_indices(wn, u0, order="blocked"):
    """Systematic resampling on the same normalised CDF and right-search as multinomial_indices:
    keys (i + u0) / N, one uniform per resampling.  Not in the reference (an option of the build,
    BASELINE north_star); this is its CPU statement for the parity test."""
    n = len(wn)
    keys = (np.arange(n, dtype=np.float64) + float(u0)) / float(n)
    return multinomial_indices(wn, keys, order=order)


def std_normal_logpdf(r):
    """scipy multivariate_normal(0, I).logpdf(r): forward_lkernel.py:35 / nuts.py:189
    with the harness' momentum proposal (run_experiments.py:111)."""
    r = np.atleast_2d(r)
    D = r.shape[1]
    return -0.5 * np.sum(r * r, axis=1) - 0.5 * D * np.log(2 * np.pi)


def gaussian_lkernel(r_new, x_new):
    """GaussianApproxLKernel.calculate_L (lkernel/gaussian_lkernel.py:24-84),
    vectorised; the per-particle scipy multivariate_normal.logpdf is restated
    through its eigh-based pseudo-inverse / pseudo-determinant (_PSD)."""
    D = x_new.shape[1]
    X = np.hstack([-r_new, x_new])
    mu_X = np.mean(X, axis=0)
    cov_X = np.cov(np.transpose(X))
    mu_r, mu_x = mu_X[:D], mu_X[D:]
    c_rr, c_rx, c_xr, c_xx = cov_X[:D, :D], cov_X[:D, D:], cov_X[D:, :D], cov_X[D:, D:]
    pinv = np.linalg.pinv(c_xx)
    cov = c_rr - c_rx @ pinv @ c_xr
    cov = cov + np.eye(D) * 1e-6
    mu = mu_r + (c_rx @ pinv @ (x_new - mu_x).T).T
    s, u = np.linalg.eigh(cov)
    eps = 1e6 * np.finfo("d").eps * np.max(np.abs(s))
    if np.min(s) < -eps:
        raise ValueError("the input matrix must be positive semidefinite")
    d = s[s > eps]
    if len(d) < len(s):
        raise np.linalg.LinAlgError("singular matrix")
    U = u * np.sqrt(1.0 / s)
    dev = (-r_new) - mu
    maha = np.sum(np.square(dev @ U), axis=1)
    return -0.5 * (D * np.log(2 * np.pi) + np.sum(np.log(d)) + maha)


def bisect_scipy(f, xa, xb, xtol=2e-12, rtol=8.881784197001252e-16, maxiter=100):
    """scipy.optimize.bisect (scipy/optimize/Zeros/bisect.c)."""
    fa, fb = f(xa), f(xb)
    if fa == 0:
        return xa
    if fb == 0:
        return xb
    if np.signbit(fa) == np.signbit(fb):
        raise ValueError("f(a) and f(b) must have different signs")
    dm = xb - xa
    for _ in range(maxiter):
        dm *= 0.5
        xm = xa + dm
        fm = f(xm)
        if fm * fa >= 0:
            xa = xm
        if fm == 0 or abs(dm) < xtol + rtol * abs(xm):
            return xm
    raise RuntimeError("Failed to converge")


def ess_tempering(N, logpri, loglik, base, phi_old, alpha=0.5):
    """ESSTempering.calculate_phi (tempering/adaptive_tempering.py:18-63)."""
    def _ess(phi):
        with np.errstate(all="ignore"):
            logw = phi * loglik + logpri - base
        index = ~np.isneginf(logw)
        ll = logsumexp_scipy(logw[index])
        with np.errstate(all="ignore"):
            wn = np.exp(logw[index] - ll)
            return 1.0 / np.sum(np.square(wn)) - N * alpha
    if _ess(1.0) >= 0:
        return 1.0
    return bisect_scipy(_ess, phi_old, 1.0)


def estimate(xc, wn):
    """Estimate._estimate (estimate/estimate.py:79-95)."""
    mean = wn.T @ xc
    var = wn.T @ np.square(xc - mean)
    return mean, var


def smc_run(target, K, N, eps, x0, logq0, lkernel="forwardsLKernel", tempering=False,
            per_iter=None, seed=0, max_depth=10, delta_max=100.0, scan_order="sequential"):
    """The loop of SMCSampler.__init__ + .sample() (smc_sampler.py:25-155) in
    the reference's order, with the NUTS proposal in C.

    per_iter: list (len K) of dicts with recorded draws {r, tape, tape_off,
    u_resample} (tape mode); None => Philox(seed) for everything.
    """
    D = target.dim
    out = dict(ess=np.zeros(K + 1), log_likelihood=np.zeros(K + 1), phi=np.zeros(K + 1),
               mean_estimate=np.zeros((K + 1, D)), variance_estimate=np.zeros((K + 1, D)),
               x_saved=np.zeros((K + 1, N, D)), logw_saved=np.zeros((K + 1, N)),
               resampled=np.zeros(K + 1, dtype=bool), nleap=np.zeros((K, N), dtype=np.int32),
               idx=[None] * K)
    x = np.array(x0, dtype=np.float64)
    phi_old = phi_new = 0.0 if tempering else 1.0

    def temper(x_new, phi_old_):
        lpri, llik = target.parts(x_new)
        base = _combine(lpri, llik, phi_old_)
        return ess_tempering(N, _combine(lpri, llik, 0.0), _combine(lpri, llik, 1.0) - _combine(lpri, llik, 0.0),
                             base, phi_old_)

    if tempering:                                   # samples.py:78-83
        phi_new = temper(x, phi_old)
        phi_old = phi_new
    logw = target.logpdf(x, phi=phi_new) - logq0    # samples.py:85
    out["x_saved"][0], out["logw_saved"][0] = x, logw

    for k in range(K):
        out["phi"][k] = phi_new
        wn, ll = normalise_weights(logw)
        mean, var = estimate(target.constrain(x), wn)
        ess = calculate_ess(wn)
        if ess < N / 2:                             # samples.py:120
            u = (per_iter[k]["u_resample"] if per_iter is not None
                 else philox_particle_uniforms(seed, k, 0, N, 2, 0))
            idx = multinomial_indices(wn, u, order=scan_order)
            x = x[idx]
            logw = np.ones(N) * ll - np.log(N)      # samples.py:143
            out["resampled"][k] = True
            out["idx"][k] = idx
        phi_used = phi_new
        if per_iter is not None:
            r = per_iter[k]["r"]
            res = nuts_rvs(target, x, r, phi_new, eps, max_depth, delta_max,
                           tape=per_iter[k]["tape"], tape_off=per_iter[k]["tape_off"])
        else:
            r = philox_normals(seed, k, N, D, 1)
            res = nuts_rvs(target, x, r, phi_new, eps, max_depth, delta_max, seed=seed, iteration=k)
        x_new, r_new = res["x_new"], res["r_new"]
        out["nleap"][k] = res["nleap"]
        if lkernel == "asymptoticLKernel":          # nuts_acc_rej.py:42-49, utils.py:3-34
            u_acc = (per_iter[k]["u_accept"] if per_iter is not None
                     else philox_particle_uniforms(seed, k, 0, N, 4, 0))
            acc = hmc_accept_reject(res, x, x_new, r, r_new, phi_used, u_acc)
            x_new = np.where(acc[:, None], x_new, x)
            r_new = np.where(acc[:, None], r_new, r)
            res["lpri1"] = np.where(acc, res["lpri1"], res["lpri0"])
            res["llik1"] = np.where(acc, res["llik1"], res["llik0"])
        if tempering:                               # samples.py:199-212
            phi_new = temper(x_new, phi_old)
        # samples.py:183-196: densities at phi = 1.0 always
        p_x = _combine(res["lpri0"], res["llik0"], 1.0)
        p_xn = _combine(res["lpri1"], res["llik1"], 1.0)
        if lkernel == "asymptoticLKernel":          # samples.py:169-180: the OLD positions, two temperatures
            logw_new = logw + _combine(res["lpri0"], res["llik0"], phi_new) - _combine(res["lpri0"], res["llik0"], phi_old)
        else:
            if lkernel == "forwardsLKernel":
                L = std_normal_logpdf(-r_new)
            elif lkernel == "GaussianApproxLKernel":
                L = gaussian_lkernel(r_new, x_new)
            else:
                raise Exception("Unknown L-kernel supplied")
            q = std_normal_logpdf(r)
            logw_new = logw + p_xn - p_x + L - q
        out["log_likelihood"][k], out["mean_estimate"][k], out["variance_estimate"][k] = ll, mean, var
        out["ess"][k] = ess
        phi_old = phi_new
        x, logw = x_new, logw_new
        out["x_saved"][k + 1], out["logw_saved"][k + 1] = x, logw
    wn, ll = normalise_weights(logw)
    mean, var = estimate(target.constrain(x), wn)
    out["ess"][K] = calculate_ess(wn)
    out["log_likelihood"][K], out["mean_estimate"][K], out["variance_estimate"][K] = ll, mean, var
    out["phi"][K] = phi_new
    if lkernel == "asymptoticLKernel":              # smc_sampler.py:152-153
        u_final = None if per_iter is None else per_iter[0].get("u_final")
        m, v = estimate_from_tempered(target, out["x_saved"], out["logw_saved"], out["phi"], u_final, seed,
                                      scan_order)
        out["mean_estimate"], out["variance_estimate"] = m, v
    return out


def hmc_accept_reject(res, x, x_new, r, r_new, phi, u):
    """proposal/utils.py:3-34 for all particles (True = accepted)."""
    with np.errstate(all="ignore"):
        H1 = _combine(res["lpri1"], res["llik1"], phi) - 0.5 * np.sum(r_new * r_new, axis=1)
        H0 = _combine(res["lpri0"], res["llik0"], phi) - 0.5 * np.sum(r * r, axis=1)
        ratio = np.exp(H1 - H0)
        prob = np.where(ratio < 1.0, ratio, 1.0)      # Python's min(1., ratio): NaN -> 1.0
        return ~((u > prob) | np.any(np.isinf(x_new), axis=1))


def estimate_from_tempered(target, x_saved, logw_saved, phi, u_final=None, seed=0, scan_order="sequential"):
    """EstimateFromTempered.estimate_from_tempered (estimate/estimate_from_tempered.py:24-55)."""
    K1, N, D = x_saved.shape
    mean, var = np.zeros((K1, D)), np.zeros((K1, D))
    for k in range(K1):
        wn, _ = normalise_weights(logw_saved[k])
        u = u_final[k] if u_final is not None else philox_particle_uniforms(seed, K1 + k, 0, N, 2, 0)
        x = x_saved[k][multinomial_indices(wn, u, order=scan_order)]
        lpri, llik = target.parts(x)
        with np.errstate(all="ignore"):
            logw = _combine(lpri, llik, 1.0) - _combine(lpri, llik, phi[k])
        ess_wn, _ = normalise_weights(logw)
        mean[k], var[k] = estimate(target.constrain(x), ess_wn)
    return mean, var


def _combine(lpri, llik, phi):
    """log pi_phi with the adapter's -inf convention (bridgestan.py:45-49)."""
    with np.errstate(all="ignore"):
        lp = lpri + phi * llik
    return np.where(np.isfinite(lp), lp, -np.inf)
