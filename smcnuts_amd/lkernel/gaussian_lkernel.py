"""Gaussian approximation of the optimal L-kernel.

Mirror of smcnuts/lkernel/gaussian_lkernel.py:4-84.  The N-scaled work (first
and second moments of X = [-r_new, x_new]; the per-particle conditional
log-density) runs on the GPU; the 2D x 2D algebra in between (pinv, eigh) is
N-independent and stays on the host in fp64 NumPy, following the reference's
own calls (np.cov, np.linalg.pinv, scipy's multivariate_normal.logpdf whose
eigh-based pseudo-inverse / pseudo-determinant is restated below) -- unless one
both covariances are comfortably positive definite: then the library does that
algebra as well (smcn_glk.hpp; several shards all-gather their moment sums and
every rank runs it) and this module only hears the status.
"""
import numpy as np

from .. import _capi


def _unpack_sums(s, E):
    """[E singles, upper triangle of the E x E products] -> (vector, symmetric matrix)."""
    v = s[:E].copy()
    Mx = np.zeros((E, E))
    iu = np.triu_indices(E)
    Mx[iu] = s[E:]
    Mx = Mx + np.triu(Mx, 1).T
    return v, Mx


class GaussianApproxLKernel:
    # D <= 32: the D x D algebra runs on the device too (smcn_gauss_lkernel_device / _stage: two Cholesky
    # factorisations by one wavefront, one host wait for the whole L-kernel instead of three).  The library refuses when a
    # covariance is not comfortably positive definite; the NumPy calls below, which ARE the reference's, then decide.
    device_algebra = True

    def __init__(self, target, N):
        self.D = target.dim
        self.N = N
        self.last_path = None        # "device" / "host": which algebra produced the last L values (tests, diagnostics)

    def conditional(self, mu_X, cov_X):
        """gaussian_lkernel.py:52-68 + scipy.stats._multivariate._PSD: returns
        (mu_x, m0, B, U, c0) with L_i = c0 - 0.5 |U^T(-r_i - m0 - B (x_i - mu_x))|^2."""
        D = self.D
        mu_r, mu_x = mu_X[:D], mu_X[D:]
        c_rr, c_rx, c_xr, c_xx = cov_X[:D, :D], cov_X[:D, D:], cov_X[D:, :D], cov_X[D:, D:]
        pinv = np.linalg.pinv(c_xx)
        cov = c_rr - c_rx @ pinv @ c_xr
        cov += np.eye(D) * 1e-6                      # :68 ridge
        B = c_rx @ pinv
        s, u = np.linalg.eigh(cov)
        eps = 1e6 * np.finfo("d").eps * np.max(np.abs(s))
        if np.min(s) < -eps:
            raise ValueError("The input matrix must be symmetric positive semidefinite.")
        d = s[s > eps]
        if len(d) < len(s):
            raise np.linalg.LinAlgError("When `allow_singular is False`, the input matrix must be "
                                        "symmetric positive definite.")
        U = u * np.sqrt(1.0 / s)
        c0 = -0.5 * (D * np.log(2 * np.pi) + np.sum(np.log(d)))
        return mu_x, mu_r, B, U, c0

    def calculate_L(self, r_new, x_new):
        """Plug-in interface on host arrays (gaussian_lkernel.py:24-84)."""
        X = np.hstack([-r_new, x_new])
        mu_x, m0, B, U, c0 = self.conditional(np.mean(X, axis=0), np.cov(np.transpose(X)))
        dev = (-r_new) - (m0 + (B @ (x_new - mu_x).T).T)
        return c0 - 0.5 * np.sum(np.square(dev @ U), axis=1)

    @staticmethod
    def _staged(ctx, comm, world, nq, n_total, info):
        """Several shards: the moment sums of every shard are all-gathered (twice: un-shifted, centred) and EVERY rank runs
        the D x D algebra on the device from the rows added in rank order -- the same bits everywhere, so every rank takes
        the same device / host decision (include/smcnuts_hip.h: smcn_gauss_lkernel_stage)."""
        import ctypes as C
        loc, gat = C.c_void_p(), C.c_void_p()
        ctx.call("smcn_gauss_lkernel_buffers", world, C.byref(loc), C.byref(gat))

        def gather():
            if world == 1:
                return
            if getattr(comm, "device_path", False):
                comm.allgather_device(loc.value, gat.value, nq)
            else:                           # communicators with a host all-gather only (gloo, tests)
                row = np.empty(nq)
                ctx.call("smcn_buf_get", loc.value, nq, _capi.dptr(row))
                rows = np.ascontiguousarray(comm.allgather(row), dtype=np.float64)
                ctx.call("smcn_buf_set", gat.value, rows.size, _capi.dptr(rows.reshape(-1)))

        ctx.call("smcn_gauss_lkernel_stage", 0, world, n_total, None)
        gather()
        ctx.call("smcn_gauss_lkernel_stage", 1, world, n_total, None)
        gather()
        ctx.call("smcn_gauss_lkernel_stage", 2, world, n_total, _capi.dptr(info))

    def apply(self, ctx, forward_kernel, comm=None, n_total=None):
        """Device path; `comm` all-gathers the shard sums."""
        D, E = self.D, 2 * self.D
        n_total = n_total or ctx.N
        nq = E + E * (E + 1) // 2
        world = 1 if comm is None else comm.world_size
        if self.device_algebra and D <= 32 and n_total >= 2:
            info = np.zeros(4)
            if world == 1 and n_total == ctx.N:
                ctx.call("smcn_gauss_lkernel_device", _capi.dptr(info))
            else:
                self._staged(ctx, comm, world, nq, float(n_total), info)
            if info[0] == 0.0:
                self.last_path = "device"
                if not forward_kernel.native_momentum:
                    r = ctx.get_proposal(x_new=False, r_new=False)[0]
                    ctx.call("smcn_set_lkernel_values", None,
                             _capi.dptr(np.ascontiguousarray(forward_kernel.logpdf(r), dtype=np.float64)))
                return _capi.LKERNEL_GAUSSIAN
        self.last_path = "host"

        def sums(shift):
            s = np.empty(nq)
            ctx.call("smcn_gauss_lkernel_sums", _capi.dptr(np.ascontiguousarray(shift, dtype=np.float64)),
                     _capi.dptr(s))
            if comm is not None and comm.world_size > 1:
                s = comm.allgather(s).sum(axis=0)
            return s

        v, _ = _unpack_sums(sums(np.zeros(E)), E)
        mu_X = v / n_total                                  # np.mean
        _, M2 = _unpack_sums(sums(mu_X), E)                 # np.cov: X -= mean; X X^T / (N - 1)
        cov_X = M2 / (n_total - 1)
        mu_x, m0, B, U, c0 = self.conditional(mu_X, cov_X)
        ctx.call("smcn_gauss_lkernel_logpdf", *(_capi.dptr(np.ascontiguousarray(a, dtype=np.float64))
                                                for a in (mu_x, m0, B, U)), float(c0))
        if not forward_kernel.native_momentum:
            r = ctx.get_proposal(x_new=False, r_new=False)[0]
            ctx.call("smcn_set_lkernel_values", None,
                     _capi.dptr(np.ascontiguousarray(forward_kernel.logpdf(r), dtype=np.float64)))
        return _capi.LKERNEL_GAUSSIAN
