"""SMC sampler with a NUTS proposal, on MI355X.

Mirror of smcnuts/smc_sampler.py:11-155 (`SMCSampler`): the constructor
SMCSampler(K, N, target, step_size, sample_proposal, momentum_proposal,
lkernel, tempering=False, rng=...) and `.sample(show_progress=True)`, then the
history attributes the reference exposes (mean_estimate, variance_estimate,
ess, phi, acceptance_rate, log_likelihood, x_saved, logw_saved, run_time).
The K-iteration loop keeps the reference's order (smc_sampler.py:109-149);
every population step is a HIP kernel over the resident particle shard.
"""
from time import time

import numpy as np

from .estimate.estimate import Estimate
from .parallel import SingleProcess
from .proposal.nuts import NUTSProposal
from .samples.samples import Samples


def _seed_from_rng(rng):
    """One 63-bit draw from the caller's NumPy RNG seeds Philox (the shared
    sequential stream of the reference cannot be consumed in parallel)."""
    if rng is None:
        return 0
    if hasattr(rng, "integers"):
        return int(rng.integers(0, 2 ** 63 - 1))
    if hasattr(rng, "randint"):
        return int(rng.randint(0, 2 ** 31 - 1)) * (2 ** 31) + int(rng.randint(0, 2 ** 31 - 1))
    return int(rng)


class SMCSampler:
    def __init__(self, K, N, target, step_size, sample_proposal=None, momentum_proposal=None,
                 lkernel="forwardsLKernel", tempering=False, rng=None, *, forward_kernel=None, verbose=False,
                 save_history=True, comm=None, device=0, seed=None, x0=None, logq0=None):
        self.K = K
        self.N = N
        self.target = target
        self.rng = rng
        self.lkernel = lkernel
        self.comm = comm or SingleProcess()
        self.verbose = verbose
        self.save_history = save_history
        if lkernel == "asymptoticLKernel":
            raise NotImplementedError("asymptoticLKernel (NUTS with accept/reject) is outside this path")
        if lkernel not in ("forwardsLKernel", "GaussianApproxLKernel"):
            raise Exception("Unknown L-kernel supplied")          # samples.py:48
        self.seed = _seed_from_rng(rng) if seed is None else int(seed)

        # smc_sampler.py:56-62 (README-style forward_kernel= overrides)
        if forward_kernel is None:
            forward_kernel = NUTSProposal(target=target, momentum_proposal=momentum_proposal,
                                          step_size=step_size, rng=rng)
        self.estimator = Estimate(target)

        # smc_sampler.py:66-74
        self.resampled = [False] * (K + 1)
        self.ess = np.zeros(K + 1)
        self.log_likelihood = np.zeros(K + 1)
        self.phi = np.zeros(K + 1)
        self.acceptance_rate = np.zeros(K + 1)
        self.leapfrogs = np.zeros(K, dtype=np.int64)
        self.run_time = None

        self.samples = Samples(N, target.dim, sample_proposal, target, forward_kernel, lkernel, tempering, rng,
                               comm=self.comm, device=device, seed=self.seed)
        self.N_local = self.samples.N_local
        self.samples.initialise_samples(x0=x0, logq0=logq0)

        Dc = getattr(target, "constrained_dim", target.dim)
        if save_history:
            self.x_saved = np.zeros([K + 1, self.N_local, target.dim])
            self.logw_saved = np.zeros([K + 1, self.N_local])
            self.x_saved[0], self.logw_saved[0], _ = self.samples.ctx.get_state()
        else:
            self.x_saved = self.logw_saved = None
        self.mean_estimate = np.zeros([K + 1, Dc])
        self.variance_estimate = np.zeros([K + 1, Dc])
        self.k = 0

    # smc_sampler.py:88-97
    def update_sampler(self, k, mean_estimate, variance_estimate, moved=0):
        self.log_likelihood[k] = self.samples.log_likelihood
        self.mean_estimate[k] = mean_estimate
        self.variance_estimate[k] = variance_estimate
        self.ess[k] = self.samples.ess
        self.acceptance_rate[k] = moved / self.N

    def step(self, tape=None, tape_off=None, r=None, u_resample=None):
        """One iteration of the loop in smc_sampler.py:109-140."""
        s, k = self.samples, self.k
        self.phi[k] = s.phi_new
        s.normalise_weights()
        mean, var = self.estimator.return_estimate_device(s.ctx, self.comm)
        s.calculate_ess()
        s.resample_if_required(u=u_resample)
        self.resampled[k] = s.resampled_last      # (the reference allocates this and never fills it)
        s.propose_samples(tape=tape, tape_off=tape_off, r=r)
        s.update_temperature()
        s.reweight()
        self.leapfrogs[k] = s.ctx.last_leapfrogs()
        moved = s.update_samples(count_moved=True)
        if self.comm.world_size > 1:
            moved = int(self.comm.allgather(np.array([float(moved)])).sum())
        self.update_sampler(k, mean, var, moved)
        if self.save_history:
            self.x_saved[k + 1], self.logw_saved[k + 1], _ = s.ctx.get_state()
        self.k += 1

    def finalise(self):
        """smc_sampler.py:143-149."""
        s = self.samples
        s.normalise_weights()
        mean, var = self.estimator.return_estimate_device(s.ctx, self.comm)
        s.calculate_ess()
        self.update_sampler(self.K, mean, var, 0)   # x is x_new after the last commit: 0, as in the reference
        self.phi[self.K] = s.phi_new

    def sample(self, show_progress=True):
        start_time = time()
        it = range(self.k, self.K)
        if show_progress:
            try:
                from tqdm import tqdm
                it = tqdm(it, desc="NUTS Sampling")
            except ImportError:
                pass
        for _ in it:
            self.step()
        self.finalise()
        self.run_time = time() - start_time
