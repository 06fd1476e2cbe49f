"""Post-hoc estimates for the asymptotic strategy.
Mirror of smcnuts/estimate/estimate_from_tempered.py:6-55 (`EstimateFromTempered`):
every saved generation is resampled from its own weights and importance-corrected
by pi(x) / pi_{phi_k}(x)."""
import numpy as np

from .. import _capi
from .estimate import device_moments
from ..parallel import combine_lse_partials
from .estimate import Estimate


class EstimateFromTempered(Estimate):
    def __init__(self, target, N, K, rng):
        super().__init__(target)
        self.N = N
        self.K = K
        self.rng = rng

    def estimate_from_tempered(self, x_saved, logw_saved, phi, ctx=None, u_final=None):
        """estimate_from_tempered.py:24-55; `ctx` is a resident shard context of N
        particles (one is created if missing); u_final[k] replays recorded uniforms."""
        K1 = x_saved.shape[0]
        own = ctx is None
        if own:
            ctx = _capi.Context(x_saved.shape[1], self.target.model_id, self.target.model_data,
                                device=getattr(self.target, "device", 0))
            if getattr(self.target, "host_evaluated", False):
                self.target.attach(ctx)
        Dc = getattr(self.target, "constrained_dim", ctx.Dc)
        mean, var = np.zeros([K1, Dc]), np.zeros([K1, Dc])
        ll, ess = np.empty(1), np.empty(1)
        for k in range(K1):
            ctx.set_state(x=x_saved[k], logw=logw_saved[k])
            ctx.call("smcn_normalise", _capi.dptr(ll), _capi.dptr(ess))                 # :38-40
            u = None if u_final is None else u_final[k]
            ctx.resample(ll[0], np.log(ctx.N), K1 + k, u=u)                              # :42-44
            ctx.call("smcn_set_logw_density_ratio", 1.0, float(phi[k]))                 # :47
            ctx.call("smcn_normalise", _capi.dptr(ll), _capi.dptr(ess))                 # :49-50
            mean[k], var[k] = device_moments(self.target, ctx)                           # :53
        if own:
            ctx.close()
        return mean, var
