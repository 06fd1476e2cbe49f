"""ESS-targeting adaptive tempering.

Mirror of smcnuts/tempering/adaptive_tempering.py:7-63 (`ESSTempering`).  The
reference evaluates the target three times per call and runs scipy's bisect on
a NumPy ESS; here the density parts (log prior, log likelihood) kept by the
NUTS kernel make every trial temperature one reduction over N on the GPU
(smcn_temper_partials), and the root bracketing is the same bisection
(scipy/optimize/Zeros/bisect.c: xtol 2e-12, rtol 4 eps, 100 iterations).
"""
import numpy as np

from ..parallel import combine_lse_partials

_XTOL, _RTOL, _MAXITER = 2e-12, 8.881784197001252e-16, 100


def bisect(f, xa, xb, xtol=_XTOL, rtol=_RTOL, maxiter=_MAXITER):
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
    raise RuntimeError("Failed to converge after %d iterations." % maxiter)


class ESSTempering:
    def __init__(self, N, target, alpha=0.5):
        self.N = N              # GLOBAL number of particles
        self.target = target
        self.alpha = alpha

    def calculate_phi_device(self, ctx, phi_old, comm=None):
        """adaptive_tempering.py:18-63 on the resident shard; the density parts
        at x_new must be on the device (kept by the NUTS kernel)."""
        def _ess(new_phi):
            p = ctx.temper_partials(phi_old, new_phi)
            parts = comm.allgather(p) if comm is not None and comm.world_size > 1 else p[None, :]
            _, sum_wn2 = combine_lse_partials(parts)
            with np.errstate(all="ignore"):
                return 1.0 / sum_wn2 - self.N * self.alpha

        if _ess(1.0) >= 0:
            return 1.0
        return bisect(_ess, phi_old, 1.0)

    def calculate_phi(self, args):
        """The reference's plug-in signature: args = [x_new, logp at phi_old, phi_old]
        on host arrays."""
        x_new, p_old, old_phi = args
        lpri, llik = self.target.logpdf_parts(x_new)

        def comb(phi):
            with np.errstate(all="ignore"):
                lp = lpri + phi * llik
            return np.where(np.isfinite(lp), lp, -np.inf)

        logpri, loglik = comb(0.0), comb(1.0) - comb(0.0)

        def _ess(new_phi):
            with np.errstate(all="ignore"):
                logw = new_phi * loglik + logpri - p_old
                logw = logw[~np.isneginf(logw)]
                mx = np.max(logw)
                wn = np.exp(logw - (mx + np.log(np.sum(np.exp(logw - mx)))))
                return 1.0 / np.sum(np.square(wn)) - self.N * self.alpha

        if _ess(1.0) >= 0:
            return 1.0
        return bisect(_ess, old_phi, 1.0)
