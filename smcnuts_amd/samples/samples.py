"""The particle set and its per-iteration population operations, on the GPU.

Mirror of smcnuts/samples/samples.py:7-222 (`Samples`): same constructor
arguments and the same method names called in the same order by SMCSampler
(normalise_weights, calculate_ess, resample_if_required, propose_samples,
update_temperature, reweight, update_samples).  State (x, r, x_new, r_new,
logw, logw_new, wn) lives on the device in [D, N] layout; the host attributes
of the same names are materialised on demand.
"""
import numpy as np

from .. import _capi
from ..lkernel.forward_lkernel import ForwardLKernel
from ..lkernel.gaussian_lkernel import GaussianApproxLKernel
from ..parallel import SingleProcess, combine_lse_partials
from ..proposal.nuts import is_standard_normal
from ..tempering.adaptive_tempering import ESSTempering


class Samples:
    def __init__(self, N, D, sample_proposal, target, forward_kernel, lkernel, tempering, rng,
                 comm=None, device=0, seed=0, shard_resampling="global", resampling="multinomial", wide_eval=True,
                 nuts_cap="auto"):
        self.comm = comm or SingleProcess()
        if shard_resampling not in ("global", "local"):
            raise ValueError("shard_resampling is 'global' or 'local'")
        self.shard_resampling = shard_resampling
        # force_exchange: run the shard protocol although there is one shard (tests of the RCCL path)
        self.sharded = self.comm.world_size > 1 or getattr(self.comm, "force_exchange", False)
        self.N = N                                   # GLOBAL number of particles
        if N % self.comm.world_size:
            raise ValueError("N must be divisible by the number of shards")
        self.N_local = N // self.comm.world_size
        self.D = D
        self.sample_proposal = sample_proposal
        self.forward_kernel = forward_kernel
        self.target = target
        self.rng = rng
        self.ctx = _capi.Context(self.N_local, target.model_id, target.model_data, device=device,
                                 particle_base=self.comm.rank * self.N_local)
        if getattr(target, "host_evaluated", False):
            target.attach(self.ctx)
        self.ctx.set_seed(seed)
        if hasattr(self.comm, "attach"):             # in-library communicator: RCCL on this context's device and stream
            try:
                self.comm.attach(self.ctx)
            except BaseException:
                self.ctx.close()                     # a failed rendezvous must not keep the shard's device buffers alive
                raise
        handle = getattr(self.comm, "stream_handle", lambda: None)()
        if handle is not None:                       # a framework communicator: launch where ITS collectives are enqueued,
            self.ctx.call("smcn_set_stream", handle)  # on every path (step-by-step included), so kernels and collectives order
        if resampling not in ("multinomial", "systematic"):
            raise ValueError("resampling is 'multinomial' (the reference's rng.choice) or 'systematic'")
        self.resampling = resampling
        self.ctx.call("smcn_set_resample_scheme", 1 if resampling == "systematic" else 0)
        # lane-per-particle kernel (arma): lane groups evaluate a wavefront's last stragglers (include/smcnuts_hip.h:
        # results then agree to rounding, not bit for bit, between differently scheduled runs); False pins the bits
        self.ctx.call("smcn_set_wide_eval", 1 if wide_eval else 0)
        # group kernels (PRMwCD): trees that want more than `doublings` doublings can be parked and finished by a second
        # launch, one wavefront per tree (include/smcnuts_hip.h: smcn_set_nuts_cap).  nuts_cap = (doublings, widen), None / 0
        # for one launch.  "auto": the target's own default -- PRMwCD with the shipped data shape parks after 9 doublings
        # and finishes the 4 % of trees that want more with the wave-per-tree evaluation (round 4, DESIGN.md 4.2: a launch
        # lasts as long as its longest tree, and a leaf of that kernel takes 2.2 us against 7.4); every other target: one launch.
        if nuts_cap == "auto":
            nuts_cap = getattr(target, "two_phase_default", None)
        # A third entry, (doublings, widen, requeue): trees that want more than `requeue` (< doublings) doublings are parked
        # there as well and taken up again by the SAME launch once the fresh particles have run out -- the launch then ends
        # on pieces of trees instead of whole ones (smcn_set_nuts_requeue; round 5: config 4 1.69 -> 1.88 G leapfrog/s).  The
        # same trees in another order: bit-identical results.
        if nuts_cap:
            d, w, rq = (nuts_cap, True, 0) if isinstance(nuts_cap, int) else (tuple(nuts_cap) + (0,))[:3]
            self.ctx.call("smcn_set_nuts_cap", int(d), 1 if w else 0)
            self.ctx.call("smcn_set_nuts_requeue", int(rq))
        self.nuts_cap = nuts_cap

        # samples.py:39-48
        if lkernel == "GaussianApproxLKernel":
            self.lkernel = GaussianApproxLKernel(target=self.target, N=self.N)
        elif lkernel == "forwardsLKernel":
            self.lkernel = ForwardLKernel(target=self.target, momentum_proposal=self.forward_kernel.momentum_proposal)
        elif lkernel == "asymptoticLKernel":
            self.lkernel = None                      # samples.py:45-46: weights from tempered densities only
        else:
            raise Exception("Unknown L-kernel supplied")

        # samples.py:51-60
        if tempering:
            self.TemperingScheme = ESSTempering(self.N, self.target, alpha=0.5)
            self.update_temperature = self._tempering
            self.phi_old = 0.0
            self.phi_new = 0.0
        else:
            self.TemperingScheme = None
            self.update_temperature = lambda: 1.0
            self.phi_old = 1.0
            self.phi_new = 1.0
        self.iteration = 0
        self.resampled_last = False
        self.total_leapfrogs = 0

    @staticmethod
    def ctx_ptr(a):
        return _capi.dptr(np.ascontiguousarray(a, dtype=np.float64))

    # ---- host views of the device state ------------------------------------------
    @property
    def x(self):
        return self.ctx.get_state(logw=False)[0]

    @property
    def logw(self):
        return self.ctx.get_state(x=False)[1]

    @property
    def wn(self):
        return self.ctx.get_state(x=False, logw=False, wn=True)[2]

    @property
    def r(self):
        return self.ctx.get_proposal(x_new=False, r_new=False)[0]

    @property
    def x_new(self):
        return self.ctx.get_proposal(r=False, r_new=False)[1]

    @property
    def r_new(self):
        return self.ctx.get_proposal(r=False, x_new=False)[2]

    @property
    def logw_new(self):
        return self.ctx.get_proposal(r=False, x_new=False, r_new=False, logw_new=True)[3]

    # ---- samples.py:63-88 ------------------------------------------------------------
    def initialise_samples(self, x0=None, logq0=None):
        native = x0 is None and (self.sample_proposal is None or is_standard_normal(self.sample_proposal, self.D))
        if not native and x0 is None:
            x0 = np.asarray(self.sample_proposal.rvs(self.N_local), dtype=np.float64).reshape(self.N_local, self.D)
        if x0 is not None:
            x0 = np.ascontiguousarray(x0, dtype=np.float64)
            if logq0 is None:
                logq0 = np.asarray(self.sample_proposal.logpdf(x0), dtype=np.float64)
            self.ctx.set_state(x=x0)
        else:
            # x ~ N(0, I) on the device; the weights are set once phi is known
            self.ctx.call("smcn_init_particles_std_normal", 1.0)
        self.ess = 0
        if self.TemperingScheme is not None:
            # samples.py:78,82: x_new = copy(x), phi_old = 0
            self.ctx.call("smcn_eval_proposed_parts", 0)
            self.phi_new = self.TemperingScheme.calculate_phi_device(self.ctx, self.phi_old, self.comm)
        else:
            self.phi_new = 1.0
        self.phi_old = self.phi_new
        lq = None if logq0 is None else np.ascontiguousarray(logq0, dtype=np.float64)
        self.ctx.call("smcn_init_weights", float(self.phi_new), _capi.dptr(lq))   # samples.py:85

    # ---- samples.py:91-113 ---------------------------------------------------------------
    def normalise_weights(self):
        p = self.ctx.normalise_partials()
        parts = self.comm.allgather(p) if self.comm.world_size > 1 else p[None, :]
        self.log_likelihood, self._sum_wn2 = combine_lse_partials(parts)
        self._local_parts = p
        self.ctx.call("smcn_normalise_apply", float(self.log_likelihood))

    def calculate_ess(self):
        with np.errstate(all="ignore"):
            self.ess = 1.0 / self._sum_wn2

    # ---- samples.py:116-146 ------------------------------------------------------------------
    def resample_if_required(self, u=None, want_idx=False):
        self.resampled_last = False
        self.last_idx = None
        if self.ess < self.N / 2:
            self._resample(u, want_idx)

    def _resample(self, u=None, want_idx=False):
        """Multinomial over the WHOLE population (samples.py:124-146).  Several shards all-gather
        weights and particles and each draws its own slice of the global ancestor indices."""
        if self.sharded and self.shard_resampling == "global":
            if u is not None or want_idx:
                raise ValueError("recorded resampling draws replay on one shard")
            self.global_resample(self.iteration, self.log_likelihood)
        elif self.comm.world_size > 1:       # "local": every shard keeps its own mass
            ll_local, _ = combine_lse_partials(self._local_parts[None, :])
            self.last_idx = self.ctx.resample(ll_local, np.log(self.N_local), self.iteration, u=u, want_idx=want_idx)
        else:
            self.last_idx = self.ctx.resample(self.log_likelihood, np.log(self.N_local), self.iteration, u=u,
                                              want_idx=want_idx)
        self.resampled_last = True

    def global_resample(self, iteration, loglik):
        """Samples._resample (samples.py:124-146) over the whole population, sharded: the result is what ONE shard
        of N particles computes (Philox keyed by global particle index).  Routed form: all-gather
        of the shards' scan-tile totals, then an all-to-all of the resampling keys and of only the ancestor rows
        each rank needs (smcn_gres_*).  Shard sizes that are no multiple of the scan tile (1024) end with a partial
        tile: the cdf is then summed in another association than one shard of N would use (ancestors can differ
        from the one-shard run only at keys within rounding of a cdf step)."""
        import ctypes as C
        comm, ctx = self.comm, self.ctx
        W, n, D = comm.world_size, ctx.N, ctx.D
        self.global_resamplings = getattr(self, "global_resamplings", 0) + 1
        device = getattr(comm, "device_path", False)
        self.global_route = "device" if device else "host"
        ll = None if loglik is None else C.byref(C.c_double(float(loglik)))
        nt = (n + 1023) // 1024          # scan tiles per shard (the last one partial when n is no multiple of 1024)
        tt = np.empty(nt)
        ctx.call("smcn_gres_begin", W, _capi.dptr(tt))
        tt_all = np.ascontiguousarray(comm.allgather(tt), dtype=np.float64).reshape(-1)     # [W * nt] doubles
        dest = np.empty(n, dtype=np.int32)
        ctx.call("smcn_gres_plan", W, comm.rank, _capi.dptr(tt_all), int(iteration), _capi.iptr(dest))
        order = np.argsort(dest, kind="stable").astype(np.int32)
        send_counts = np.bincount(dest, minlength=W).astype(np.int64)
        counts = np.ascontiguousarray(comm.allgather(send_counts.astype(np.float64))).astype(np.int64)   # [src][dst]
        recv_counts = np.ascontiguousarray(counts[:, comm.rank])
        m = int(recv_counts.sum())
        ctx.call("smcn_gres_set_order", _capi.iptr(order))
        ctx.call("smcn_gres_reserve", m)
        ptrs = [C.c_void_p() for _ in range(6)]
        ctx.call("smcn_gres_buffers", *(C.byref(p) for p in ptrs))
        _, _, keys_send, keys_recv, rows_send, rows_recv = (p.value for p in ptrs)
        xchg = getattr(comm, "exchange", None)
        if xchg is None:
            from ..parallel import exchange_through_host
            xchg = lambda *a: exchange_through_host(comm, *a)
        xchg(ctx, keys_send, send_counts, keys_recv, recv_counts, 1)
        ctx.call("smcn_gres_serve", W, comm.rank, m)
        xchg(ctx, rows_send, recv_counts, rows_recv, send_counts, D)
        ctx.call("smcn_gres_finish", W, ll)
        self.rows_moved = getattr(self, "rows_moved", 0) + int(send_counts.sum() - send_counts[comm.rank])

    # ---- samples.py:149-158 ---------------------------------------------------------------------
    def propose_samples(self, tape=None, tape_off=None, r=None, u_accept=None):
        if u_accept is not None:
            self.forward_kernel.propose(self.ctx, self.phi_new, self.iteration, tape=tape, tape_off=tape_off, r=r,
                                        u_accept=u_accept)
        else:
            self.forward_kernel.propose(self.ctx, self.phi_new, self.iteration, tape=tape, tape_off=tape_off, r=r)

    # ---- samples.py:161-196 ----------------------------------------------------------------------
    def reweight(self):
        if self.lkernel is None:                     # samples.py:169-180 (_asymptotic_reweight)
            self.ctx.call("smcn_reweight_asymptotic", float(self.phi_old), float(self.phi_new))
            return
        if isinstance(self.lkernel, GaussianApproxLKernel):
            code = self.lkernel.apply(self.ctx, self.forward_kernel, self.comm, self.N)
        else:
            code = self.lkernel.apply(self.ctx, self.forward_kernel)
        self.ctx.call("smcn_reweight", code)

    # ---- samples.py:199-212 -------------------------------------------------------------------------
    def _tempering(self):
        self.phi_new = self.TemperingScheme.calculate_phi_device(self.ctx, self.phi_old, self.comm)
        return self.phi_new

    # ---- samples.py:215-222 (+ the acceptance statistic of smc_sampler.py:97) ---------------------
    def update_samples(self, count_moved=True):
        self.phi_old = self.phi_new
        moved = self.ctx.commit(count_moved)
        self.iteration += 1
        return moved
