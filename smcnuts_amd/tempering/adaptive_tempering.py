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

    device_bisection = True      # False: one reduction + host wait per trial temperature (the round-2 path; A/B, tests)

    def calculate_phi_device(self, ctx, phi_old, comm=None):
        """adaptive_tempering.py:18-63 on the resident shard; the density parts
        at x_new must be on the device (kept by the NUTS kernel).  The bisection itself runs on the device
        (smcn_temper_bisect*: bisect.c's iteration, four steps per pass): one host synchronisation per call; shards whose
        communicator all-gathers device memory exchange 60 doubles per pass in the stream.  Communicators that only
        offer a host all-gather keep the host-driven bisection below."""
        import ctypes as C
        world = 1 if comm is None else comm.world_size
        sharded = comm is not None and (world > 1 or getattr(comm, "force_exchange", False))
        target = self.N * self.alpha
        if self.device_bisection and (not sharded or getattr(comm, "device_path", False)):
            phi, status = C.c_double(0.0), C.c_int(1)
            if not sharded:
                ctx.call("smcn_temper_bisect", float(phi_old), float(target), C.byref(phi), C.byref(status))
            else:
                loc, gat = C.c_void_p(), C.c_void_p()
                ctx.call("smcn_temper_bisect_buffers", world, C.byref(loc), C.byref(gat))
                p, upto = 0, 1                  # the opening alone first (ESS(1) >= target: phi = 1), then the bisection whole
                while True:
                    while p < upto:
                        ctx.call("smcn_temper_bisect_pass", p, float(phi_old))
                        if world > 1:
                            comm.allgather_device(loc.value, gat.value, 60)
                        ctx.call("smcn_temper_bisect_decide", p, world, float(target), float(phi_old))
                        p += 1
                    ctx.call("smcn_temper_bisect_result", C.byref(phi), C.byref(status))
                    if status.value != 1 or p >= 26:
                        break
                    upto = 11 if upto == 1 else min(26, upto + 4)
            if status.value == 2:
                raise ValueError("f(a) and f(b) must have different signs")
            if status.value != 0:
                raise RuntimeError("Failed to converge after 100 iterations.")
            return float(phi.value)

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
