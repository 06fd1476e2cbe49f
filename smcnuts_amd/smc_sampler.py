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
from .estimate.estimate_from_tempered import EstimateFromTempered
from .proposal.nuts_acc_rej import NUTSProposalWithAccRej
from .parallel import SingleProcess
from .proposal.nuts import NUTSProposal
from .samples.samples import Samples


def _seed_from_rng(rng):
    """One 63-bit draw from the caller's NumPy RNG seeds Philox (the shared
    sequential stream of the reference cannot be consumed in parallel)."""
    if rng is None:
        # the reference's default is a fresh np.random.default_rng() (smc_sampler.py:35): seeded from the system's entropy
        return int(np.random.SeedSequence().generate_state(2, dtype=np.uint32).astype(np.uint64) @ np.array([1, 2 ** 31], dtype=np.uint64))
    if hasattr(rng, "integers"):
        return int(rng.integers(0, 2 ** 63 - 1))
    if hasattr(rng, "randint"):
        return int(rng.randint(0, 2 ** 31 - 1)) * (2 ** 31) + int(rng.randint(0, 2 ** 31 - 1))
    return int(rng)


def block_size_from_ess(ess_seen, N, B, fmax):
    """Iterations to speculate on next, from the last two ESS values of clean generations:
    geometric extrapolation of the decay to the resampling threshold N/2 (samples.py:120) with a
    20 % margin; at least double the block just validated, at most fmax."""
    grow = min(2 * B, fmax)
    if len(ess_seen) == 2 and ess_seen[0] > 0 and ess_seen[1] > 0:
        a, b = float(ess_seen[0]), float(ess_seen[1])
        if b >= a:                                   # not decaying
            grow = fmax if b > 0.5 * N else grow
        elif b > 0.5 * N:
            left = np.log(0.5 * N / b) / np.log(b / a)
            grow = max(grow, int(min(fmax, 0.8 * left)))
    return max(1, min(grow, fmax))


class SMCSampler:
    """SMCSampler(K, N, target, step_size, sample_proposal, momentum_proposal, lkernel, tempering, rng) as in
    smcnuts/smc_sampler.py:25-36, plus keyword-only extensions.  `wide_eval` (default True; recorded in `self.wide_eval`):
    the arma kernel lets idle lanes of a wavefront share its stragglers' recurrences, which re-associates likelihood sums
    depending on the launch schedule -- run-to-run results are bit-identical, but fused blocks against one launch per
    iteration, or 8 shards against one, agree to rounding only.  wide_eval=False is the reproducible mode: every
    evaluation by one lane, identical bits under every schedule, shard count and block size (about 16 % slower).
    `preallocate` (default True): the constructor allocates every device buffer of the device-resident loop -- the history of
    all K + 1 generations, block partials, per-transition records -- so that sample() starts with its first launch; a caller
    that only drives the host loop (step() / finalise(), which read none of them) or that is short of device memory passes
    preallocate=False and the buffers are made by the first step_async() / run_fused() / sample() instead."""

    def __init__(self, K, N, target, step_size, sample_proposal=None, momentum_proposal=None,
                 lkernel="forwardsLKernel", tempering=False, rng=None, *, forward_kernel=None, verbose=False,
                 save_history=True, comm=None, device=0, seed=None, x0=None, logq0=None,
                 shard_resampling="global", resampling="multinomial", wide_eval=True, nuts_cap="auto", preallocate=True):
        from .model.targets import as_target
        target = as_target(target)      # host-evaluated targets are wrapped
        self.K = K
        self.N = N
        self.target = target
        self.rng = rng
        self.lkernel = lkernel
        self.comm = comm or SingleProcess()
        self.verbose = verbose
        self.save_history = save_history
        self.wide_eval = bool(wide_eval)
        if lkernel not in ("forwardsLKernel", "GaussianApproxLKernel", "asymptoticLKernel"):
            raise Exception("Unknown L-kernel supplied")          # samples.py:48
        if seed is None and rng is None and self.comm.world_size > 1:
            raise ValueError("several shards need ONE seed: pass seed= (or an identically seeded rng=) on every rank")
        self.seed = _seed_from_rng(rng) if seed is None else int(seed)

        # smc_sampler.py:56-62 (README-style forward_kernel= overrides)
        if forward_kernel is None:
            cls = NUTSProposalWithAccRej if lkernel == "asymptoticLKernel" else NUTSProposal   # smc_sampler.py:45-60
            forward_kernel = cls(target=target, momentum_proposal=momentum_proposal, step_size=step_size, rng=rng)
        self.estimator = (EstimateFromTempered(target, N, K, rng) if lkernel == "asymptoticLKernel"
                          else Estimate(target))
        self.estimator.seed = self.seed

        # smc_sampler.py:66-74
        self.resampled = [False] * (K + 1)
        self.ess = np.zeros(K + 1)
        self.log_likelihood = np.zeros(K + 1)
        self.phi = np.zeros(K + 1)
        self.acceptance_rate = np.zeros(K + 1)
        self.leapfrogs = np.zeros(K, dtype=np.int64)
        self.run_time = None
        self.speculate = True            # fused blocks: enqueue the next block before the flags are read
        self.discarded_launches = 0

        self.samples = Samples(N, target.dim, sample_proposal, target, forward_kernel, lkernel, tempering, rng,
                               comm=self.comm, device=device, seed=self.seed, shard_resampling=shard_resampling, resampling=resampling,
                               wide_eval=wide_eval, nuts_cap=nuts_cap)
        self.N_local = self.samples.N_local
        self.samples.initialise_samples(x0=x0, logq0=logq0)

        Dc = getattr(target, "constrained_dim", target.dim)
        self._touch = None
        self._dl_upto = 0                # generations <= this one are already in x_saved / logw_saved
        self._overlap_history = False    # sample(): download validated generations beside the loop (smcn_history_download)
        if save_history:
            # smc_sampler.py:73-74.  The pages are TOUCHED before the history is downloaded into them (a copy that faults
            # 100 MB in page by page takes five times as long) -- by a thread, beside the construction and the first
            # launches, not in the caller's way (round 3: np.full here, 7 of the constructor's 17 ms)
            self.x_saved = np.zeros([K + 1, self.N_local, target.dim])
            self.logw_saved = np.zeros([K + 1, self.N_local])
        else:
            self.x_saved = self.logw_saved = None
        self.mean_estimate = np.zeros([K + 1, Dc])
        self.variance_estimate = np.zeros([K + 1, Dc])
        self.k = 0
        # Device-resident loop: forward L-kernel at a fixed temperature with the N(0, I)
        # momentum proposal (BASELINE configs 1-3, 5).  Gaussian L-kernel / tempering need
        # host algebra (pinv/eigh, bisection) between kernels and run step by step.
        self.device_resident = (lkernel == "forwardsLKernel" and not tempering
                                and getattr(forward_kernel, "native_momentum", False)
                                and not getattr(target, "host_evaluated", False))
        self._fast_started = False
        self._host_loop_used = False
        if save_history:
            # (measured and kept, profiles/r05_stall_trace.txt: this pageable copy is the call that stalls for 16-18 ms in the
            #  second sampler of a process, one run out of three -- but leaving generation 0 to the device history instead made
            #  the FIRST cold sample() 8-10 ms slower: its first big download then is the process's first pageable copy into
            #  these pages, 10.3 ms for 84 MB instead of 2.5)
            self.x_saved[0], self.logw_saved[0], _ = self.samples.ctx.get_state()
        if self.device_resident and preallocate:
            # every device buffer of the loop NOW (history, block partials, transition records): a cold sample() then
            # starts with its first launch instead of with allocations
            self._fast_start()
            if self.samples.ctx.fused_transitions:
                self._fuse_setup(32)
        if save_history and self.x_saved[1:].nbytes > (8 << 20):
            import threading

            # Four threads, a quarter of the pages each, in generation order: zeroing 134 MB of fresh pages takes one core about
            # as long as the whole run (9.5 ms), and the first block's rows are asked for after ~2 ms -- on a busy host one
            # thread did not keep ahead of the downloads (run_time 15-20 ms instead of 9.5 in two of four runs of round 5)
            flat = (self.x_saved[1:].reshape(-1), self.logw_saved[1:].reshape(-1))

            def touch(part, parts=4):
                for a in flat:
                    n = a.size
                    lo, hi = (n * part) // parts, (n * (part + 1)) // parts
                    a[lo:hi:512] = 0.0       # one write per 4 KB page (NumPy releases the GIL for the loop)

            class _Touchers:
                def __init__(self, ths):
                    self.ths = ths

                def join(self):
                    for t in self.ths:
                        t.join()

                def is_alive(self):
                    return any(t.is_alive() for t in self.ths)

            # (started AFTER the device allocations above: page faults and hipMalloc contend for the process's memory map)
            ths = [threading.Thread(target=touch, args=(i,), daemon=True) for i in range(4)]
            for t in ths:
                t.start()
            self._touch = _Touchers(ths)

    # smc_sampler.py:88-97
    def update_sampler(self, k, mean_estimate, variance_estimate, moved=0):
        self.log_likelihood[k] = self.samples.log_likelihood
        self.mean_estimate[k] = mean_estimate
        self.variance_estimate[k] = variance_estimate
        self.ess[k] = self.samples.ess
        self.acceptance_rate[k] = moved / self.N

    def step(self, tape=None, tape_off=None, r=None, u_resample=None, u_accept=None):
        """One iteration of the loop in smc_sampler.py:109-140."""
        s, k = self.samples, self.k
        self._host_loop_used = True
        self.phi[k] = s.phi_new
        s.normalise_weights()
        mean, var = self.estimator.return_estimate_device(s.ctx, self.comm)
        s.calculate_ess()
        s.resample_if_required(u=u_resample)
        self.resampled[k] = s.resampled_last      # (the reference allocates this and never fills it)
        s.propose_samples(tape=tape, tape_off=tape_off, r=r, u_accept=u_accept)
        s.update_temperature()
        s.reweight()
        self.leapfrogs[k] = s.ctx.last_leapfrogs()
        moved = s.update_samples(count_moved=True)
        if self.comm.world_size > 1:
            moved = int(self.comm.allgather(np.array([float(moved)])).sum())
        self.update_sampler(k, mean, var, moved)
        if self.save_history:
            self._history_ready()
            self.x_saved[k + 1], self.logw_saved[k + 1], _ = s.ctx.get_state()
        self.k += 1

    def finalise(self, u_final=None):
        """smc_sampler.py:143-153."""
        s = self.samples
        s.normalise_weights()
        mean, var = self.estimator.return_estimate_device(s.ctx, self.comm)
        s.calculate_ess()
        self.update_sampler(self.K, mean, var, 0)   # x is x_new after the last commit: 0, as in the reference
        self.phi[self.K] = s.phi_new
        if self.lkernel == "asymptoticLKernel":     # smc_sampler.py:152-153
            if not self.save_history:
                raise RuntimeError("asymptoticLKernel estimates need save_history=True")
            self.mean_estimate, self.variance_estimate = self.estimator.estimate_from_tempered(
                self.x_saved, self.logw_saved, self.phi, u_final=u_final,
                samples=s if s.sharded else None)

    # ---- device-resident variant of step()/finalise() --------------------------------
    def _fast_start(self):
        s = self.samples
        s.ctx.fast_begin(self.K, self.save_history, self.comm.world_size)
        self._fast_started = True                    # (the context already launches on the communicator's stream: Samples.__init__)

    def _exchange(self):
        """The one exchange of an iteration: all-gather of 4 + 2*Dc shard partials."""
        c, comm = self.samples.ctx, self.comm
        if comm.world_size == 1 and not getattr(comm, "force_exchange", False):
            return
        if getattr(comm, "device_path", False):
            comm.allgather_device(c.lp_ptr, c.gath_ptr, c.nq)
        else:
            c.partials_set_gathered(comm.allgather(c.partials_get()))

    def step_async(self, tape=None, tape_off=None, r=None, u_resample=None):
        """smc_sampler.py:109-140, enqueued without waiting for the device."""
        if not self.device_resident:
            raise RuntimeError("this configuration runs step by step (use step())")
        if self._host_loop_used:
            raise RuntimeError("step_async() cannot follow step()")
        if self.samples.sharded:         # several shards: global resampling lives in the block driver
            if tape is not None or r is not None or u_resample is not None:
                raise ValueError("recorded draws replay on one shard")
            return self.run_fused(upto=self.k + 1, fuse_max=1)
        if not self._fast_started:
            self._fast_start()
        s, fk, k = self.samples, self.samples.forward_kernel, self.k
        if r is not None:
            s.ctx.call("smcn_set_momentum", s.ctx_ptr(r))
        if u_resample is not None:
            s.ctx.call("smcn_set_resample_uniforms", s.ctx_ptr(u_resample))
        s.ctx.step_begin(k)
        self._exchange()
        s.ctx.step_finish(k, self.comm.world_size, self.comm.rank, self.N, fk.step_size, s.phi_new, fk.max_depth,
                          fk.delta_max, False, tape, tape_off)
        s.iteration += 1
        self.k += 1

    def _history_ready(self):
        t = self._touch
        if t is not None:
            t.join()
            self._touch = None

    def _download_validated(self, k):
        """Generations up to k are final (their blocks have been waited for): their rows go to x_saved / logw_saved now, on
        the library's download stream, under the NUTS launch that is already enqueued."""
        if not (self.save_history and self._overlap_history) or k <= self._dl_upto:
            return
        if self._touch is not None and self._touch.is_alive():
            return          # the pages are still being touched: these rows go with a later block's (or with the rest, at the end)
        self._history_ready()
        self.samples.ctx.call("smcn_history_download", self._dl_upto + 1, int(k), self.samples.ctx_ptr(self.x_saved),
                              self.samples.ctx_ptr(self.logw_saved))
        self._dl_upto = int(k)

    def download_history(self):
        """x_saved / logw_saved (smc_sampler.py:73-74,139-140) from the device history."""
        if self.save_history and self._fast_started and not self._host_loop_used:
            self._history_ready()
            _, self.x_saved, self.logw_saved = self.samples.ctx.fast_read(self.K, True, self.x_saved, self.logw_saved,
                                                                           k_from=self._dl_upto + 1)
            self._dl_upto = self.K

    def finalise_async(self, download_history=True):
        """smc_sampler.py:143-149 on the device, then ONE synchronisation and download."""
        s, K = self.samples, self.K
        known = getattr(self, "_known_flag", None)
        if not (known is not None and known[0] == K and self.k == K):
            # generation K's scalars (the pipelined block driver has already produced them with the block's statistics)
            s.ctx.step_begin(K)
            self._exchange()
            s.ctx.step_finish(K, self.comm.world_size, self.comm.rank, self.N, 0.0, s.phi_new, last=True)
        want_hist = self.save_history and download_history
        if want_hist:
            self._history_ready()
        hist, xs, lw = s.ctx.fast_read(K, want_hist, self.x_saved, self.logw_saved,
                                       k_from=(self._dl_upto + 1) if want_hist else 0)
        if want_hist:
            self._dl_upto = K
        Dc = self.mean_estimate.shape[1]
        self.log_likelihood[:] = hist[:, 0]
        self.ess[:] = hist[:, 1]
        self.resampled = [bool(v) for v in hist[:K, 2]] + [False]   # generation K is only estimated (smc_sampler.py:143-149)
        self.leapfrogs[:] = hist[:K, 3].astype(np.int64)
        moved = hist[:, 4].copy()
        if self.comm.world_size > 1:
            moved = self.comm.allgather(moved).sum(axis=0)
        self.acceptance_rate[:] = moved / self.N
        self.acceptance_rate[K] = 0.0
        self.phi[:] = hist[:, 5]
        self.mean_estimate[:] = hist[:, 6:6 + Dc]
        self.variance_estimate[:] = hist[:, 6 + Dc:6 + 2 * Dc]
        s.log_likelihood, s.ess = self.log_likelihood[K], self.ess[K]
        if self.save_history and download_history:
            self.x_saved, self.logw_saved = xs, lw

    # ---- fused transitions: up to `fuse_max` iterations per NUTS launch -------------------
    def run_fused(self, upto=None, fuse_max=32):
        """Advance to iteration `upto` (default K) with several iterations per NUTS launch.
        Between resampling events a particle's next transition depends only on its own
        sample, so B iterations run inside one launch, speculating that none of the
        generations in between falls below the resampling threshold (samples.py:120).  The
        library checks the speculation on the recorded weights and rolls back to the first
        generation that has to resample, so the results equal step_async()'s -- bit for bit with
        wide_eval=False; with the default (lane groups evaluate a wavefront's stragglers: WHICH
        evaluations depends on the launch's schedule) the same trees and states to rounding;
        B adapts to the decay of the ESS (block_size_from_ess), back to 1 after a roll-back."""
        import ctypes as C
        if not self.device_resident:
            raise RuntimeError("this configuration runs step by step (use step())")
        s, fk, comm = self.samples, self.samples.forward_kernel, self.comm
        ctx = s.ctx
        upto = self.K if upto is None else min(int(upto), self.K)
        if self._host_loop_used:
            raise RuntimeError("run_fused() cannot follow step()")
        if not self._fast_started:
            self._fast_start()
        self._fuse_setup(fuse_max)
        if self.samples.ctx.fused_transitions:
            return self._run_blocks(upto)
        while self.k < upto:       # models without the fused-transition kernel: one iteration per block
            ctx.step_begin(self.k)
            self._exchange()
            decided = 0
            if s.sharded and s.shard_resampling == "global":   # decision + resampling over all shards
                flag = C.c_int(0)
                ctx.call("smcn_fuse_decide", self.k, comm.world_size, comm.rank, float(self.N), float(s.phi_new),
                         C.byref(flag))
                if flag.value:
                    self._global_resample()
                decided = 1
            ctx.call("smcn_fuse_run", self.k, 1, comm.world_size, comm.rank, float(self.N), float(fk.step_size),
                     float(s.phi_new), fk.max_depth, fk.delta_max, decided)
            n_ok = C.c_int(0)
            ctx.call("smcn_fuse_finish", self.k, 1, comm.world_size, comm.rank, float(self.N), float(s.phi_new),
                     C.byref(n_ok))
            self.k += n_ok.value
            s.iteration += n_ok.value
            self._advance_bar()

    def _fuse_setup(self, fuse_max):
        """Buffers for blocks of up to fuse_max iterations per NUTS launch (grown, never shrunk)."""
        import ctypes as C
        ctx, comm = self.samples.ctx, self.comm
        # the kernel addresses its per-transition records with 32-bit byte offsets
        rec_bytes = 8 * (2 * ((ctx.D + 1) & ~1) + 6)
        fuse_max = int(max(1, min(fuse_max, 64, (2 ** 32 - 1) // (ctx.N * rec_bytes), max(self.K, 1))))
        if getattr(self, "_fuse_max", 0) < fuse_max:
            if getattr(self, "_fuse_max", 0) > 0:
                ctx.call("smcn_synchronize")
            ctx.call("smcn_fuse_begin", int(fuse_max), comm.world_size)
            a, b, n = C.c_void_p(), C.c_void_p(), C.c_int()
            ctx.call("smcn_fuse_buffers", C.byref(a), C.byref(b), C.byref(n))
            self._fuse_lp, self._fuse_gath, self._fuse_max, self._fuse_B = a.value, b.value, fuse_max, 1

    def _run_blocks(self, upto):
        """Pipelined fused blocks.  The statistics of a block are enqueued behind its NUTS launch;
        before waiting for them the NEXT block is enqueued from the block's last generation,
        speculating that nothing in between has to resample (true for all but a handful of
        iterations of a run), so the device does not idle while the host looks at the flags.  A
        failed speculation discards that launch and restarts from the generation that resamples."""
        import ctypes as C
        s, fk, comm, ctx = self.samples, self.samples.forward_kernel, self.comm, self.samples.ctx
        W, rank, Nf, phi = comm.world_size, comm.rank, float(self.N), float(s.phi_new)
        eps, md, dm = float(fk.step_size), fk.max_depth, fk.delta_max
        fmax = self._fuse_max

        def launch(k0, B):
            ctx.call("smcn_block_launch", k0, B, eps, phi, md, dm)

        def exchange(B):
            if W == 1 and not getattr(comm, "force_exchange", False):
                return
            if getattr(comm, "device_path", False):
                comm.allgather_device(self._fuse_lp, self._fuse_gath, B * ctx.nq)
            else:
                p = np.empty(B * ctx.nq)
                ctx.call("smcn_block_partials_get", B, s.ctx_ptr(p))
                g = np.ascontiguousarray(comm.allgather(p))
                ctx.call("smcn_block_partials_set", B, W, s.ctx_ptr(g))

        k = self.k
        known = getattr(self, "_known_flag", None)
        known = known if (known is not None and known[0] == k) else None
        inflight = None
        while k < upto:
            if inflight is None:
                if known is None or known[1]:
                    # generation k's scalars (and resampling) the classic way
                    ctx.step_begin(k)
                    self._exchange()
                    flag = C.c_int(0)
                    ctx.call("smcn_fuse_decide", k, W, rank, Nf, phi, C.byref(flag))
                    if flag.value:
                        if s.sharded and s.shard_resampling == "global":
                            s.global_resample(k, None)
                        else:
                            ctx.call("smcn_block_resample_local", k)
                B = self._cut_final_block(max(1, min(self._fuse_B, upto - k, fmax)), upto - k)
                launch(k, B)
                inflight = (k, B)
            k0, B = inflight
            ctx.call("smcn_block_post", k0, B, W)
            exchange(B)
            ctx.call("smcn_block_stats", k0, B, W, rank, Nf, phi, int(k0 + B >= self.K))
            nxt = None
            if k0 + B < upto and self.speculate:
                B2 = self._cut_final_block(max(1, min(max(2 * B, getattr(self, "_spec_hint", 1)), fmax, upto - (k0 + B))),
                                           upto - (k0 + B))
                ctx.call("smcn_block_commit", k0, B)
                launch(k0 + B, B2)
                nxt = (k0 + B, B2)
            else:
                ctx.call("smcn_block_commit", k0, B)      # speculative too: redone below if the block is cut short
            n_ok, res = C.c_int(0), C.c_int(0)
            ctx.call("smcn_block_wait", B, C.byref(n_ok), C.byref(res))
            ok = n_ok.value
            if ok == B and not res.value:
                k, inflight, known = k0 + B, nxt, (k0 + B, 0)
                self._fuse_B = self._spec_hint = self._next_block_size(B, fmax)
                self._download_validated(k)       # (the next block is already running)
                if getattr(self, "_bar", None) is not None:
                    self._bar.update(k - self._bar.n)
            else:
                if nxt is not None:
                    ctx.call("smcn_synchronize")      # the speculative launch is discarded
                    self.discarded_launches += 1
                ctx.call("smcn_block_commit", k0, ok)
                self._download_validated(k0 + ok)
                k, inflight, known = k0 + ok, None, (k0 + ok, 1)
                self._fuse_B = 1 if ok < B else min(2 * B, fmax)
                self._spec_hint, self._ess_seen = 1, []
        self._known_flag = known
        s.iteration += k - self.k
        self.k = k

    # ---- checkpoint / restore of the device-resident loop (a generation boundary) ------------
    def checkpoint(self):
        """Particle state and loop bookkeeping at the current generation (device-resident loop).
        The reference keeps no checkpoints; bench.py uses this to time the same K
        iterations several times from one saved state."""
        s = self.samples
        s.ctx.call("smcn_synchronize")
        x, logw, _ = s.ctx.get_state()
        return dict(k=self.k, iteration=s.iteration, x=x, logw=logw, known=getattr(self, "_known_flag", None),
                    fuse_B=getattr(self, "_fuse_B", 1), spec=getattr(self, "_spec_hint", 1),
                    ess_seen=list(getattr(self, "_ess_seen", [])), discarded=self.discarded_launches)

    def restore(self, ck):
        s = self.samples
        s.ctx.call("smcn_synchronize")
        s.ctx.set_state(x=ck["x"], logw=ck["logw"])
        self.k, s.iteration = ck["k"], ck["iteration"]
        self._known_flag, self._fuse_B, self._spec_hint = ck["known"], ck["fuse_B"], ck["spec"]
        self._ess_seen, self.discarded_launches = list(ck["ess_seen"]), ck["discarded"]
        self._dl_upto = min(self._dl_upto, self.k)

    def _cut_final_block(self, B, left):
        """With the history downloaded beside the loop, only the LAST block's rows travel after the device has gone idle: a
        block that would finish the run is cut at 60 % so that the rest (its rows: 40 % of the block's) is all that is left
        exposed.  (No history / no overlap: one block, as long as the speculation allows.)"""
        if self.save_history and self._overlap_history and B >= left and left >= 12:
            return int(np.ceil(0.6 * left))
        return B

    def _next_block_size(self, B, fmax):
        ess = np.empty(B)
        self.samples.ctx.call("smcn_block_ess", B, self.samples.ctx_ptr(ess))
        self._ess_seen = (getattr(self, "_ess_seen", []) + list(ess))[-2:]
        return block_size_from_ess(self._ess_seen, self.N, B, fmax)

    def _global_resample(self):
        self.samples.global_resample(self.k, None)

    def _advance_bar(self):
        bar = getattr(self, "_bar", None)
        if bar is not None and self.k > bar.n:
            bar.update(self.k - bar.n)

    def sample(self, show_progress=True):
        start_time = time()
        if self.device_resident and not self._host_loop_used:
            # smc_sampler.py:109: the bar advances by validated generations (a fused block at a time)
            self._bar = None
            if show_progress and self.comm.rank == 0:
                try:
                    from tqdm import tqdm
                    self._bar = tqdm(total=self.K, initial=self.k, desc="NUTS Sampling")
                except ImportError:
                    pass
            self._overlap_history = bool(self.save_history)
            try:
                if self.samples.ctx.fused_transitions or self.samples.sharded:
                    self.run_fused()
                else:
                    for _ in range(self.k, self.K):
                        self.step_async()
                        self._advance_bar()
                self.finalise_async()
                self._advance_bar()
            finally:
                self._overlap_history = False
                if self._bar is not None:
                    self._bar.close()
                    self._bar = None
            self.run_time = time() - start_time
            return
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
