"""Particle sharding: the shard partials combine to the global
log-likelihood / ESS on every rank, over a real torch.distributed group
(gloo, world_size 2, CPU) and -- on the GPU -- two shards of one sampler give
the single-shard results."""
import os
import socket
import threading

import numpy as np
import pytest

from _tol import close

from oracle import oracle as orc
from smcnuts_amd.parallel import SingleProcess, combine_lse_partials


def lse_partials_np(logw):
    """What smcn_normalise_partials returns for a shard (host restatement for the test)."""
    a = logw[~np.isneginf(logw)]
    if a.size == 0:
        return np.array([-np.inf, 0.0, 0.0, 0.0])
    mx = a.max()
    shift = mx if np.isfinite(mx) else 0.0
    e = np.exp(a - shift)
    return np.array([mx, float(np.sum(a == mx)), float(np.sum(e[a != mx])), float(np.sum(e * e))])


def test_combine_matches_reference_normalisation():
    rng = np.random.default_rng(0)
    for trial in range(20):
        logw = rng.normal(size=1000) * 5
        if trial % 3 == 0:
            logw[rng.integers(0, 1000, 50)] = -np.inf
        if trial % 4 == 0:
            logw[:3] = logw.max()          # ties at the maximum
        wn, ll = orc.normalise_weights(logw)
        for nshard in (1, 2, 8):
            parts = [lse_partials_np(s) for s in np.split(logw, nshard)]
            cl, swn2 = combine_lse_partials(np.array(parts))
            close(cl, ll, rtol=1e-13)
            close(1.0 / swn2, orc.calculate_ess(wn), rtol=1e-11)
    # a shard whose weights are all -inf contributes nothing
    parts = [lse_partials_np(np.full(10, -np.inf)), lse_partials_np(np.array([0.0, 1.0]))]
    cl, _ = combine_lse_partials(np.array(parts))
    close(cl, np.log(1 + np.e), rtol=1e-14)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from smcnuts_amd.parallel import TorchDistComm
    comm = TorchDistComm()
    logw = np.random.default_rng(123).normal(size=4096) * 4          # same on every rank
    shard = np.split(logw, world)[rank]
    parts = comm.allgather(lse_partials_np(shard))
    ll, swn2 = combine_lse_partials(parts)
    # moments: all-gather of shard sums, added in rank order
    x = np.random.default_rng(5).normal(size=(4096, 3))
    wn = np.exp(logw - ll)
    xs, ws = np.split(x, world)[rank], np.split(wn, world)[rank]
    mean = comm.allgather(ws @ xs).sum(axis=0)
    q.put((rank, ll, swn2, mean))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_allgather_combine():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    logw = np.random.default_rng(123).normal(size=4096) * 4
    wn, ll = orc.normalise_weights(logw)
    for r in res:
        close(r[1], ll, rtol=1e-13)
        close(1.0 / r[2], orc.calculate_ess(wn), rtol=1e-11)
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]          # bit-identical on both ranks
    np.testing.assert_array_equal(res[0][3], res[1][3])
    close(res[0][3], wn @ np.random.default_rng(5).normal(size=(4096, 3)), rtol=1e-12)


class ThreadComm:
    """In-process stand-in for a 2-rank communicator (two shards on one GPU)."""

    def __init__(self, world):
        self.world_size = world
        self._bar = threading.Barrier(world)
        self._slots = [None] * world
        self._local = threading.local()

    def bind(self, rank):
        self._local.rank = rank
        return self

    @property
    def rank(self):
        return self._local.rank

    def allgather(self, v):
        self._slots[self.rank] = np.array(v, dtype=np.float64)
        self._bar.wait()
        out = np.stack(self._slots)
        self._bar.wait()
        return out


@pytest.mark.gpu
@pytest.mark.parametrize("lkernel", ["forwardsLKernel", "GaussianApproxLKernel"])
def test_two_shards_equal_one_shard_until_resampling(lkernel):
    """Philox is keyed by the global particle index, so two shards of N/2 draw
    what one shard of N draws; all global scalars agree to reduction round-off.
    (Gaussian target at this step size never resamples, so local == global.)"""
    from smcnuts_amd import GaussianTarget, SMCSampler
    K, N, seed = 5, 8192, 31
    one = SMCSampler(K=K, N=N, target=GaussianTarget(4), step_size=0.1, lkernel=lkernel, seed=seed)
    one.sample(show_progress=False)       # forward L-kernel: the device-resident loop, host exchange between shards
    assert not any(one.resampled)
    comm = ThreadComm(2)
    out = [None, None]

    class RankView:
        def __init__(self, r):
            self.r = r
            self.world_size = 2
        rank = property(lambda self: self.r)
        def allgather(self, v):
            comm.bind(self.r)
            return comm.allgather(v)

    def run(r):
        s = SMCSampler(K=K, N=N, target=GaussianTarget(4), step_size=0.1, lkernel=lkernel, seed=seed,
                       comm=RankView(r))
        s.sample(show_progress=False)
        out[r] = s

    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    a, b = out
    np.testing.assert_array_equal(a.ess, b.ess)
    np.testing.assert_array_equal(a.mean_estimate, b.mean_estimate)
    close(a.ess, one.ess, rtol=1e-13)
    close(a.log_likelihood, one.log_likelihood, rtol=1e-14, atol=1e-15)
    close(a.mean_estimate, one.mean_estimate, rtol=1e-12, atol=1e-15)
    close(a.variance_estimate, one.variance_estimate, rtol=1e-12, atol=1e-15)
    close(np.concatenate([a.x_saved, b.x_saved], axis=1), one.x_saved, rtol=1e-9 if lkernel != "forwardsLKernel" else 0, atol=0)
    close(np.concatenate([a.logw_saved, b.logw_saved], axis=1), one.logw_saved, rtol=1e-12, atol=1e-12)
    assert a.leapfrogs.sum() + b.leapfrogs.sum() == one.leapfrogs.sum()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["global", "local"])
def test_two_shards_fused_equals_two_shards_stepwise(mode):
    """Sharded + fused transitions (host exchange of the (B-1) x nq block) reproduce the
    sharded one-iteration-per-launch loop bit for bit, including (global) resampling."""
    from smcnuts_amd import ArmaModel, SMCSampler
    K, N, seed = 12, 4096, 21

    def run_pair(fused):
        comm = ThreadComm(2)
        out = [None, None]

        class RankView:
            def __init__(self, r):
                self.r, self.world_size = r, 2
            rank = property(lambda self: self.r)
            def allgather(self, v):
                comm.bind(self.r)
                return comm.allgather(v)

        def run(r):
            s = SMCSampler(K=K, N=N, target=ArmaModel(), step_size=0.01, seed=seed, comm=RankView(r), wide_eval=False,
                           shard_resampling=mode)
            if fused:
                s.run_fused(fuse_max=4)
            else:
                for _ in range(K):
                    s.step_async()
            s.finalise_async()
            out[r] = s

        th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
        return out

    a, b = run_pair(False), run_pair(True)
    for r in range(2):
        assert a[r].resampled == b[r].resampled and any(a[r].resampled)
        np.testing.assert_array_equal(a[r].x_saved, b[r].x_saved)
        np.testing.assert_array_equal(a[r].logw_saved, b[r].logw_saved)
        np.testing.assert_array_equal(a[r].leapfrogs, b[r].leapfrogs)
        close(a[r].ess, b[r].ess, rtol=1e-14)
        close(a[r].mean_estimate, b[r].mean_estimate, rtol=1e-14, atol=1e-16)
    np.testing.assert_array_equal(b[0].ess, b[1].ess)        # global scalars identical on both shards
    if mode == "local":      # shard masses never mix: the degenerate first generation pins ESS below N_local
        assert all(a[0].resampled[:-1]) and a[0].ess.max() <= N // 2 + 1


def _run_shards(make, world, drive, device=False):
    """`world` samplers of one process, one thread each.  device=False: the communicator offers a host all-gather only
    (every exchange goes through NumPy); device=True: smcnuts_amd.parallel.InProcessComm, whose all-gather and all-to-all
    move DEVICE buffers -- the code path of RcclComm / TorchDistComm("nccl")."""
    out = [None] * world
    errs = []
    if device:
        from smcnuts_amd.parallel import InProcessComm
        group = InProcessComm(world)
        views = [group.view(r) for r in range(world)]
    else:
        comm = ThreadComm(world)

        class RankView:
            def __init__(self, r):
                self.r, self.world_size = r, world
            rank = property(lambda self: self.r)
            def allgather(self, v):
                comm.bind(self.r)
                return comm.allgather(v)

        views = [RankView(r) for r in range(world)]

    def run(r):
        try:
            s = make(views[r])
            drive(s)
            out[r] = s
        except BaseException as e:          # a dead rank must not leave the others waiting at a barrier for ever
            errs.append((r, e))
            bar = group._bar if device else comm._bar
            bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=900)
    real = [e for e in errs if not isinstance(e[1], threading.BrokenBarrierError)] or errs
    if real:
        raise real[0][1]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("world,fuse_max,scheme", [(2, 1, "multinomial"), (2, 4, "multinomial"), (4, 4, "multinomial"),
                                                   (2, 4, "systematic")])
def test_shards_resample_globally_like_one_shard(world, fuse_max, scheme):
    """arma from N(0, I): the first generations are degenerate (ESS of a few particles) and
    resample.  Resampling is global -- all-gather + the ancestor indices one shard of N particles
    draws -- so the sharded run IS the unsharded run: same resampling decisions, same particles."""
    from smcnuts_amd import ArmaModel, SMCSampler
    K, N, seed = 10, 4096, 21
    one = SMCSampler(K=K, N=N, target=ArmaModel(), step_size=0.01, seed=seed, resampling=scheme, wide_eval=False)
    one.sample(show_progress=False)
    assert any(one.resampled) and not all(one.resampled)

    def drive(s):
        s.run_fused(fuse_max=fuse_max)
        s.finalise_async()

    sh = _run_shards(lambda c: SMCSampler(K=K, N=N, target=ArmaModel(), step_size=0.01, seed=seed, comm=c, wide_eval=False,
                                          resampling=scheme), world, drive)
    for s in sh:
        assert s.resampled == one.resampled
        close(s.ess, one.ess, rtol=1e-12)
        close(s.log_likelihood, one.log_likelihood, rtol=1e-14, atol=2e-15)
        close(s.mean_estimate, one.mean_estimate, rtol=1e-12, atol=1e-15)
        close(s.variance_estimate, one.variance_estimate, rtol=5e-11, atol=5e-15)
        close(s.acceptance_rate, one.acceptance_rate, rtol=0, atol=1e-15)
    np.testing.assert_array_equal(np.concatenate([s.x_saved for s in sh], axis=1), one.x_saved)
    assert sum(int(s.leapfrogs.sum()) for s in sh) == int(one.leapfrogs.sum())


@pytest.mark.gpu
def test_shards_stepwise_path_resamples_globally():
    """The step-by-step loop (Gaussian-approximation L-kernel) with two shards: global resampling
    through the host all-gather; same decisions and estimates as one shard."""
    from smcnuts_amd import ArmaModel, SMCSampler
    K, N, seed = 6, 2048, 5
    kw = dict(K=K, N=N, target=None, step_size=0.01, seed=seed, lkernel="GaussianApproxLKernel", wide_eval=False)
    one = SMCSampler(**{**kw, "target": ArmaModel()})
    one.sample(show_progress=False)
    assert any(one.resampled)
    sh = _run_shards(lambda c: SMCSampler(**{**kw, "target": ArmaModel()}, comm=c), 2,
                     lambda s: s.sample(show_progress=False))
    for s in sh:
        assert list(s.resampled) == list(one.resampled)
        assert s.samples.lkernel.last_path == "device"      # moment sums all-gathered through the host, the D x D algebra on every rank's GPU
        close(s.ess, one.ess, rtol=1e-10)
        close(s.mean_estimate, one.mean_estimate, rtol=1e-10, atol=1e-13)
    close(np.concatenate([s.x_saved for s in sh], axis=1), one.x_saved, rtol=1e-10, atol=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("world,N", [(2, 2048), (2, 3000)])
def test_asymptotic_strategy_on_shards_equals_one_shard(world, N):
    """asymptoticLKernel + tempering with the population split over shards: the per-particle Metropolis test is
    keyed by the global particle index and estimate_from_tempered (estimate_from_tempered.py:24-55) normalises,
    resamples and averages over the whole population, so the shards reproduce the one-shard run
    (N = 3000: shard sizes that are no multiple of the scan tile end with a partial tile in the routed resampling)."""
    from smcnuts_amd import ArmaModel, SMCSampler
    K, seed = 5, 9
    kw = dict(K=K, N=N, step_size=0.01, seed=seed, lkernel="asymptoticLKernel", tempering=True, wide_eval=False)
    one = SMCSampler(target=ArmaModel(), **kw)
    one.sample(show_progress=False)
    sh = _run_shards(lambda c: SMCSampler(target=ArmaModel(), comm=c, **kw), world,
                     lambda s: s.sample(show_progress=False))
    for s in sh:
        assert list(s.resampled) == list(one.resampled)
        close(s.phi, one.phi, rtol=1e-12)
        close(s.ess, one.ess, rtol=1e-11)
        close(s.acceptance_rate, one.acceptance_rate, rtol=0, atol=1e-15)
        close(s.mean_estimate, one.mean_estimate, rtol=1e-11, atol=1e-13)
        close(s.variance_estimate, one.variance_estimate, rtol=1e-10, atol=1e-13)
    close(np.concatenate([s.x_saved for s in sh], axis=1), one.x_saved, rtol=1e-11, atol=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("world,N", [(2, 2048), (2, 3000)])
def test_wide_particles_on_shards_equal_one_shard(world, N):
    """D = 200 (wave-per-particle kernel, per-XCD queues, row gather, the kernel's own re-weighting statistics) with the
    population split over shards and a target that makes the first generations degenerate: global resampling moves
    200-double rows between shards (N = 3000: shards ending in a partial scan tile) and the run is the one-shard run."""
    from smcnuts_amd import GaussianTarget, SMCSampler
    K, seed, D = 4, 3, 200
    mk = lambda: GaussianTarget(D, prior_sd=3.0, lik_mean=0.3, lik_sd=0.8)
    kw = dict(K=K, N=N, step_size=0.05, seed=seed)
    one = SMCSampler(target=mk(), **kw)
    one.sample(show_progress=False)
    assert any(one.resampled)

    def drive(s):
        s.run_fused(fuse_max=4)
        s.finalise_async()

    sh = _run_shards(lambda c: SMCSampler(target=mk(), comm=c, **kw), world, drive)
    for s in sh:
        assert s.resampled == one.resampled
        close(s.ess, one.ess, rtol=1e-12)
        close(s.log_likelihood, one.log_likelihood, rtol=1e-14, atol=1e-12)
        close(s.mean_estimate, one.mean_estimate, rtol=1e-11, atol=1e-13)
        close(s.acceptance_rate, one.acceptance_rate, rtol=0, atol=1e-15)
    np.testing.assert_array_equal(np.concatenate([s.x_saved for s in sh], axis=1), one.x_saved)
    assert sum(int(s.leapfrogs.sum()) for s in sh) == int(one.leapfrogs.sum())


@pytest.mark.gpu
@pytest.mark.parametrize("world,N", [(2, 4096), (2, 3000)])
def test_a_shard_that_serves_no_ancestors(world, N):
    """Degenerate first generation: every particle of the LAST shard starts where the target's density is -inf-like
    (weights vanish against the first shard's), so all ancestors live on shard 0 and shard 1 serves ZERO requests of the
    routed resampling (smcn_gres_reserve(0) still owns buffers; exchanges with an empty side complete on every rank).
    The sharded run is the one-shard run."""
    from smcnuts_amd import GaussianTarget, SMCSampler
    K, seed, D = 3, 17, 4
    rng = np.random.default_rng(seed)
    x0 = rng.normal(size=(N, D))
    x0[N // 2:] += 60.0                      # log-weights ~ -1800 below the first half's: wn underflows to exactly 0
    logq0 = np.zeros(N)
    kw = dict(K=K, N=N, target=None, step_size=0.2, seed=seed)
    one = SMCSampler(**{**kw, "target": GaussianTarget(D)}, x0=x0, logq0=logq0)
    one.sample(show_progress=False)
    assert one.resampled[0]
    moved = []

    def drive(s):
        s.sample(show_progress=False)
        moved.append((s.comm.rank, getattr(s.samples, "rows_moved", 0)))

    sh = _run_shards(lambda c: SMCSampler(**{**kw, "target": GaussianTarget(D)}, comm=c,
                                          x0=np.split(x0, world)[c.rank], logq0=np.split(logq0, world)[c.rank]), world, drive)
    for s in sh:
        assert list(s.resampled) == list(one.resampled)
        close(s.ess, one.ess, rtol=1e-12)
        close(s.mean_estimate, one.mean_estimate, rtol=1e-11, atol=1e-13)
    np.testing.assert_array_equal(np.concatenate([s.x_saved for s in sh], axis=1), one.x_saved)
    assert dict(moved)[world - 1] >= N // world          # the last shard fetched every ancestor from shard 0


@pytest.mark.gpu
@pytest.mark.parametrize("model,lkernel,world,N", [("arma", "forwardsLKernel", 2, 4096), ("arma", "GaussianApproxLKernel", 4, 4096),
                                                   ("prmwcd", "GaussianApproxLKernel", 2, 2048), ("arma", "forwardsLKernel", 2, 3000)])
def test_device_exchange_between_shards_tempering_lkernel_resampling(model, lkernel, world, N):
    """The shard protocol ON DEVICE BUFFERS (what RcclComm runs over xGMI), rehearsed by in-process shards whose all-gather and
    all-to-all are device-to-device copies: the ESS bisection of the tempering across shards (smcn_temper_bisect_pass ->
    all-gather of 60 doubles -> smcn_temper_bisect_decide reading the gathered rows), the Gaussian L-kernel's staged moment
    sums with the D x D algebra on every rank, and the routed global resampling (keys and ancestor rows point to point;
    N = 3000: shards ending in a partial scan tile).  phi, ESS, estimates and particles are the one-shard run's."""
    from smcnuts_amd import ArmaModel, PRMwCDModel, SMCSampler
    mk = ArmaModel if model == "arma" else PRMwCDModel
    K = 6 if model == "arma" else 4
    kw = dict(K=K, N=N, step_size=0.01, seed=13, lkernel=lkernel, tempering=True, wide_eval=False)
    one = SMCSampler(target=mk(), **kw)
    one.sample(show_progress=False)
    assert any(one.resampled) and 0 < one.phi[0] < 1
    used = []

    def drive(s):
        s.sample(show_progress=False)
        used.append((s.comm.rank, dict(s.comm.device_calls), getattr(s.samples, "global_route", None),
                     getattr(s.samples.lkernel, "last_path", None)))

    sh = _run_shards(lambda c: SMCSampler(target=mk(), comm=c, **kw), world, drive, device=True)
    for rank, calls, route, path in used:
        assert calls["allgather"] > K and calls["exchange"] >= 2 and route == "device"
        if lkernel == "GaussianApproxLKernel":
            assert path == "device"
    chaotic = model == "prmwcd"          # PRMwCD trajectories amplify the last bits of the tempering ladder
    # (Gaussian L-kernel: the moment sums of the shards are added in another association than one shard's -- its L values,
    # hence the weights, agree to ~1e-9 rather than to the last bits)
    tol = 1e-10 if lkernel == "GaussianApproxLKernel" else 1e-12      # (observed: 2e-15 .. 2e-14)
    for s in sh:
        assert list(s.resampled) == list(one.resampled)
        close(s.phi, one.phi, rtol=tol if not chaotic else 1e-6)
        close(s.ess, one.ess, rtol=10 * tol if not chaotic else 0.2)
        close(s.acceptance_rate, one.acceptance_rate, rtol=0, atol=1e-12 if not chaotic else 0.05)
        if not chaotic:
            close(s.mean_estimate, one.mean_estimate, rtol=10 * tol, atol=1e-10)
    for a in sh[1:]:                    # bit-identical scalars on every rank
        np.testing.assert_array_equal(a.phi, sh[0].phi)
        np.testing.assert_array_equal(a.ess, sh[0].ess)
        np.testing.assert_array_equal(a.mean_estimate, sh[0].mean_estimate)
    if not chaotic:
        close(np.concatenate([s.x_saved for s in sh], axis=1), one.x_saved, rtol=10 * tol, atol=1e-10)


@pytest.mark.gpu
def test_device_exchange_of_fused_blocks_equals_host_exchange():
    """Fused blocks over shards: the batched partials of B generations all-gathered on the device (InProcessComm) give the
    run the host all-gather gives, bit for bit, and both are the one-shard run."""
    from smcnuts_amd import ArmaModel, SMCSampler
    K, N, seed = 12, 4096, 21
    kw = dict(K=K, N=N, step_size=0.01, seed=seed, wide_eval=False)

    def drive(s):
        s.run_fused(fuse_max=4)
        s.finalise_async()

    host = _run_shards(lambda c: SMCSampler(target=ArmaModel(), comm=c, **kw), 2, drive, device=False)
    dev = _run_shards(lambda c: SMCSampler(target=ArmaModel(), comm=c, **kw), 2, drive, device=True)
    for a, b in zip(host, dev):
        assert a.resampled == b.resampled and any(a.resampled)
        np.testing.assert_array_equal(a.x_saved, b.x_saved)
        np.testing.assert_array_equal(a.logw_saved, b.logw_saved)
        np.testing.assert_array_equal(a.ess, b.ess)
        np.testing.assert_array_equal(a.mean_estimate, b.mean_estimate)
        assert b.comm.device_calls["allgather"] > 0 and b.samples.global_route == "device"
