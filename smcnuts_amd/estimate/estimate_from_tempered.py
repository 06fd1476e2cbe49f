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

    def estimate_from_tempered(self, x_saved, logw_saved, phi, ctx=None, u_final=None, samples=None):
        """estimate_from_tempered.py:24-55; `ctx` is a resident shard context of N
        particles (one is created if missing); u_final[k] replays recorded uniforms.
        Several shards pass their `samples` (the rows of x_saved / logw_saved are this shard's):
        normalisation, the unconditional resampling of every generation and the moments then run
        over the whole population, as one shard of N particles computes them."""
        K1 = x_saved.shape[0]
        if samples is not None and samples.sharded:
            if u_final is not None:
                raise ValueError("recorded resampling draws replay on one shard")
            return self._from_tempered_sharded(x_saved, logw_saved, phi, samples)
        own = ctx is None
        if own:
            ctx = _capi.Context(x_saved.shape[1], self.target.model_id, self.target.model_data,
                                device=getattr(self.target, "device", 0))
            if getattr(self.target, "host_evaluated", False):
                self.target.attach(ctx)
            ctx.set_seed(getattr(self, "seed", 0))       # the sampler's Philox seed (SMCSampler sets it)
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

    def _from_tempered_sharded(self, x_saved, logw_saved, phi, samples):
        ctx, comm = samples.ctx, samples.comm
        K1 = x_saved.shape[0]
        Dc = getattr(self.target, "constrained_dim", ctx.Dc)
        mean, var = np.zeros([K1, Dc]), np.zeros([K1, Dc])

        def total(s):
            return comm.allgather(s).sum(axis=0) if comm.world_size > 1 else s
        keep = samples.shard_resampling
        samples.shard_resampling = "global"          # (the estimate is a statement about the whole population)
        x_end, logw_end, _ = ctx.get_state()         # the sampler's final state is put back afterwards:
        ll_end, swn2_end, ess_end = samples.log_likelihood, getattr(samples, "_sum_wn2", None), samples.ess   # particles, weights, scalars
        try:
            for k in range(K1):
                ctx.set_state(x=x_saved[k], logw=logw_saved[k])
                samples.normalise_weights()                                              # :38-40
                samples.global_resample(K1 + k, samples.log_likelihood)                  # :42-44
                ctx.call("smcn_set_logw_density_ratio", 1.0, float(phi[k]))              # :47
                samples.normalise_weights()                                              # :49-50
                mean[k], var[k] = device_moments(self.target, ctx, total)                # :53
        finally:
            samples.shard_resampling = keep
            ctx.set_state(x=x_end, logw=logw_end)
            samples.normalise_weights()              # the device's wn is the final generation's again ...
            samples.log_likelihood, samples.ess = ll_end, ess_end          # ... and the host scalars exactly what sample() left
            if swn2_end is not None:
                samples._sum_wn2 = swn2_end
        return mean, var
