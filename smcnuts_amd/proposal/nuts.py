"""NUTS as an SMC proposal, on the GPU.

Mirror of smcnuts/proposal/nuts.py:6-189 (`NUTSProposal`): same constructor
(target, momentum_proposal, step_size, rng) and the operator interface
`rvs(x_cond, r_cond, phi) -> (x_prime, r_prime)`, `logpdf(r)`.  The reference
loops over particles in Python and recurses per tree; here every particle's
transition runs in one kernel launch (smcnuts_amd/csrc/smcn_nuts.hpp).
"""
import numpy as np

from .. import _capi

MAX_TREE_DEPTH = 10   # nuts.py:4
DELTA_MAX = 100.0     # nuts.py:125


def is_standard_normal(dist, dim):
    """True for scipy's frozen multivariate_normal(0, I) -- the momentum /
    sample proposal of experiments/run_experiments.py:110-111."""
    mean, cov = getattr(dist, "mean", None), getattr(dist, "cov", None)
    if mean is None or cov is None or callable(mean) or callable(cov):
        return False
    mean, cov = np.atleast_1d(np.asarray(mean, dtype=float)), np.atleast_2d(np.asarray(cov, dtype=float))
    return (mean.shape == (dim,) and cov.shape == (dim, dim) and not mean.any()
            and np.array_equal(cov, np.eye(dim)))


class NUTSProposal:
    def __init__(self, target, momentum_proposal, step_size, rng=None, max_depth=MAX_TREE_DEPTH,
                 delta_max=DELTA_MAX):
        from ..model.targets import as_target
        self.target = target = as_target(target)     # host-evaluated targets are wrapped
        self.momentum_proposal = momentum_proposal
        self.step_size = step_size
        self.rng = rng
        self.max_depth = int(max_depth)
        self.delta_max = float(delta_max)
        self.native_momentum = momentum_proposal is None or is_standard_normal(momentum_proposal, target.dim)
        self._ctx = None
        self._calls = 0
        self.last_stats = None

    # ---- fast path used by Samples: state stays on the device -----------------
    def propose(self, ctx, phi, iteration, tape=None, tape_off=None, r=None):
        """Samples.propose_samples (samples.py:149-158) on a resident shard."""
        if r is not None:
            ctx.call("smcn_set_momentum", _capi.dptr(np.ascontiguousarray(r, dtype=np.float64)))
        elif not self.native_momentum:
            rr = np.ascontiguousarray(self.momentum_proposal.rvs(ctx.N), dtype=np.float64).reshape(ctx.N, ctx.D)
            ctx.call("smcn_set_momentum", _capi.dptr(rr))
        ctx.propose_nuts(self.step_size, phi, iteration, self.max_depth, self.delta_max, tape, tape_off)

    # ---- the reference's operator interface (host arrays in/out) ---------------
    def rvs(self, x_cond, r_cond, phi=1.0, tape=None, tape_off=None, seed=None):
        """nuts.py:34-56.  RNG: recorded tapes (exact replay of the reference's
        draws) or Philox keyed by (seed, call number, particle)."""
        x_cond = np.ascontiguousarray(x_cond, dtype=np.float64)
        r_cond = np.ascontiguousarray(r_cond, dtype=np.float64)
        N = x_cond.shape[0]
        if self._ctx is None or self._ctx.N != N:
            if self._ctx is not None:
                self._ctx.close()
            self._ctx = _capi.Context(N, self.target.model_id, self.target.model_data,
                                      device=getattr(self.target, "device", 0))
            if getattr(self.target, "host_evaluated", False):
                self.target.attach(self._ctx)
        c = self._ctx
        if seed is not None:
            c.set_seed(seed)
        c.set_state(x=x_cond)
        c.call("smcn_set_momentum", _capi.dptr(r_cond))
        c.propose_nuts(self.step_size, phi, self._calls, self.max_depth, self.delta_max, tape, tape_off)
        self._calls += 1
        _, x_new, r_new, _ = c.get_proposal(r=False)
        self.last_stats = c.tree_stats()
        self.last_stats.update(zip(("lpri0", "llik0", "lpri1", "llik1"), c.density_parts()))
        return x_new, r_new

    def logpdf(self, r):
        """nuts.py:177-189: log-density of the forward kernel = the momentum proposal."""
        if self.momentum_proposal is not None:
            return self.momentum_proposal.logpdf(r)
        r = np.atleast_2d(r)
        return -0.5 * np.sum(r * r, axis=1) - 0.5 * r.shape[1] * np.log(2 * np.pi)
