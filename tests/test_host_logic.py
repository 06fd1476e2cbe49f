"""Host-side pieces of the path that need no GPU: the bisection of the ESS tempering, the 2D x 2D
algebra of the Gaussian L-kernel, the accept test of the asymptotic strategy, the estimator, the
block-size / speculation bookkeeping of the pipelined driver.  Checked against SciPy (the
reference's own dependencies) and the oracle."""
import os

import numpy as np
import pytest
import scipy.optimize
from scipy.stats import multivariate_normal

from oracle import oracle as orc


def test_bisect_is_scipys_bisect():
    """tempering/adaptive_tempering.py:bisect restates scipy/optimize/Zeros/bisect.c (the call at
    smcnuts/tempering/adaptive_tempering.py:63): same iterates, same return value."""
    from smcnuts_amd.tempering.adaptive_tempering import bisect
    fs = [(lambda x: x * x - 0.3, 0.0, 1.0), (lambda x: np.cos(3 * x) - 0.2, 0.0, 1.0),
          (lambda x: np.exp(-5 * x) - 0.5, 0.01, 1.0), (lambda x: x - 1.0, 0.0, 1.0), (lambda x: x, 0.0, 1.0)]
    for f, a, b in fs:
        assert bisect(f, a, b) == scipy.optimize.bisect(f, a, b)
        assert bisect(f, a, b) == orc.bisect_scipy(f, a, b)
    with pytest.raises(ValueError):
        bisect(lambda x: x + 1.0, 0.0, 1.0)


class _T:
    dim = 3


def test_gaussian_lkernel_host_algebra_matches_per_particle_scipy():
    """lkernel/gaussian_lkernel.py:calculate_L against the reference's formulation evaluated
    literally (gaussian_lkernel.py:52-84: np.cov, pinv, one scipy multivariate_normal per particle)."""
    from smcnuts_amd.lkernel.gaussian_lkernel import GaussianApproxLKernel
    rng = np.random.default_rng(3)
    N, D = 200, 3
    A = rng.standard_normal((D, D))
    x = rng.standard_normal((N, D)) @ A
    r = 0.5 * x + rng.standard_normal((N, D))
    L = GaussianApproxLKernel(_T(), N).calculate_L(r, x)
    X = np.hstack([-r, x])
    mu, cov = np.mean(X, axis=0), np.cov(X.T)
    pinv = np.linalg.pinv(cov[D:, D:])
    C = cov[:D, :D] - cov[:D, D:] @ pinv @ cov[D:, :D] + 1e-6 * np.eye(D)
    want = np.array([multivariate_normal(mean=mu[:D] + cov[:D, D:] @ pinv @ (x[i] - mu[D:]), cov=C).logpdf(-r[i])
                     for i in range(N)])
    np.testing.assert_allclose(L, want, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(L, orc.gaussian_lkernel(r, x), rtol=1e-10, atol=1e-10)


def test_gaussian_lkernel_singular_covariance_raises_like_scipy():
    from smcnuts_amd.lkernel.gaussian_lkernel import GaussianApproxLKernel
    k = GaussianApproxLKernel(_T(), 10)
    cov = np.zeros((6, 6))
    cov[3:, 3:] = np.eye(3)
    cov[:3, :3] = -np.eye(3)          # negative conditional covariance: not PSD
    with pytest.raises(ValueError):
        k.conditional(np.zeros(6), cov)


def test_accept_mask_follows_python_min_semantics():
    """proposal/utils.py: rejected iff u > min(1, exp(dH)) or an infinite coordinate; NaN accepts."""
    from smcnuts_amd.proposal.utils import hmc_accept_mask
    rng = np.random.default_rng(0)
    N, D = 64, 4
    x_new = rng.standard_normal((N, D))
    r, r_new = rng.standard_normal((N, D)), rng.standard_normal((N, D))
    lp0, lp1 = rng.standard_normal(N), rng.standard_normal(N)
    lp1[3] = np.nan
    x_new[5, 2] = np.inf
    lp1[7] = -np.inf
    u = rng.uniform(size=N)
    keep = hmc_accept_mask(lp0, lp1, r, r_new, x_new, u)
    for i in range(N):
        with np.errstate(all="ignore"):
            ratio = np.exp((lp1[i] - 0.5 * r_new[i] @ r_new[i]) - (lp0[i] - 0.5 * r[i] @ r[i]))
        rejected = (u[i] > min(1.0, ratio)) or np.isinf(x_new[i]).any()
        assert keep[i] == (not rejected)
    assert keep[3] and not keep[5] and not keep[7]


def test_estimator_plugin_signature():
    from smcnuts_amd.estimate.estimate import Estimate

    class T:
        constrained_dim = 2
        @staticmethod
        def constrain(x):
            y = np.array(x, copy=True)
            y[:, 1] = np.exp(y[:, 1])
            return y

    rng = np.random.default_rng(1)
    x = rng.standard_normal((50, 2))
    wn = rng.uniform(size=50)
    wn /= wn.sum()
    mean, var = Estimate(T()).return_estimate(x, wn)
    xc = T.constrain(x)
    np.testing.assert_allclose(mean, (wn[:, None] * xc).sum(0), rtol=1e-14)
    np.testing.assert_allclose(var, (wn[:, None] * (xc - mean) ** 2).sum(0), rtol=1e-13)
    m2, v2 = orc.estimate(xc, wn)
    np.testing.assert_allclose(mean, m2, rtol=1e-14)
    np.testing.assert_allclose(var, v2, rtol=1e-13)


def test_systematic_keys_properties():
    """oracle.systematic_indices: sorted ancestors, counts within one of N w (any u0)."""
    rng = np.random.default_rng(5)
    for n in (1, 7, 1000):
        w = rng.uniform(size=n) ** 4
        w /= w.sum()
        for u0 in (0.0, 0.37, 0.999999):
            idx = orc.systematic_indices(w, u0)
            assert np.all(np.diff(idx) >= 0) and idx.min() >= 0 and idx.max() < n
            assert np.all(np.abs(np.bincount(idx, minlength=n) - n * w) < 1 + 1e-9)


def test_host_targets_still_need_the_gpu_library():
    """A host-evaluated target only moves the DENSITY to the caller: without a GPU the
    sampler fails loudly at context creation (no CPU path); an object without the StanModel surface
    is refused up front."""
    from smcnuts_amd import SMCSampler
    import torch

    class HostTarget:
        dim = 2
        def logpdf(self, x, phi=1.0):
            return -0.5 * np.sum(np.square(x), axis=-1)
        def logpdfgrad(self, x, phi=1.0):
            return -np.asarray(x)

    with pytest.raises(TypeError):
        SMCSampler(K=2, N=8, target=object(), step_size=0.1)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            SMCSampler(K=2, N=8, target=HostTarget(), step_size=0.1)


def test_block_size_policy():
    """smc_sampler.block_size_from_ess: never below doubling, never above the cap, the whole cap
    when the ESS is flat, about 80 % of the extrapolated distance to N/2 otherwise."""
    from smcnuts_amd.smc_sampler import block_size_from_ess as f
    N = 65536
    assert f([], N, 1, 64) == 2 and f([100.0], N, 4, 64) == 8
    assert f([60000.0, 60000.0], N, 2, 64) == 64                  # flat
    assert f([65000.0, 64805.0], N, 4, 64) == 64                  # 0.3 % per iteration: ~228 left
    left = np.log(0.5 * N / 40000.0) / np.log(40000.0 / 42000.0)  # ~4.1 iterations left
    assert f([42000.0, 40000.0], N, 1, 64) == max(2, int(0.8 * left))
    assert f([42000.0, 40000.0], N, 8, 64) == 16                  # doubling still wins
    assert f([30000.0, 29000.0], N, 2, 64) == 4                   # already below the threshold: plain doubling
    assert all(1 <= f([a, b], N, B, 16) <= 16 for a in (1.0, 5e4) for b in (1.0, 4e4, 6e4) for B in (1, 8, 16))


def test_rccl_rendezvous_hands_the_id_to_every_rank(tmp_path, monkeypatch):
    """RcclComm's rendezvous (the RCCL id from rank 0 to the other ranks of the launch, through a file keyed by
    the launcher) with a stand-in for the library: every rank ends with rank 0's 128 bytes; a second communicator
    of the same launch uses another key."""
    import threading
    from smcnuts_amd import _capi, parallel

    class FakeLib:
        calls = 0

        def smcn_comm_unique_id(self, buf):
            FakeLib.calls += 1
            buf.raw = bytes([FakeLib.calls]) * 128
            return 0

    monkeypatch.setattr(_capi, "lib", lambda: FakeLib())
    monkeypatch.setenv("SMCN_RENDEZVOUS_DIR", str(tmp_path))
    for round_ in (1, 2):
        got = {}

        def run(rank):
            c = parallel.RcclComm(rank=rank, world_size=4, addr="127.0.0.1", port=12345, tag=round_)
            got[rank] = c._share_id()

        th = [threading.Thread(target=run, args=(r,)) for r in (3, 1, 2)]
        for t in th:
            t.start()
        run(0)
        for t in th:
            t.join(timeout=30)
        assert set(got) == {0, 1, 2, 3} and all(v == bytes([round_]) * 128 for v in got.values())
    assert len(list(tmp_path.iterdir())) == 2


def test_rccl_rendezvous_decides_by_content_not_by_clock(tmp_path, monkeypatch):
    """A rank accepts the id file only when it carries THIS launch's nonce (launcher pid + its start time) behind the 128
    id bytes -- however long ago rank 0 wrote it (a staggered start must not lose a valid id: ADVICE r04) -- and never the
    file another launch left under the same key."""
    import os
    import time
    from smcnuts_amd import parallel
    monkeypatch.setenv("SMCN_RENDEZVOUS_DIR", str(tmp_path))
    monkeypatch.setenv("SMCN_RENDEZVOUS_TIMEOUT", "0.4")
    c = parallel.RcclComm(rank=1, world_size=2, addr="127.0.0.1", port=23456, tag=7)
    path = c._id_path()
    with open(path, "wb") as f:                       # an earlier launch's file: right size, another nonce
        f.write(b"\x07" * 128 + b"1:12345")
    with pytest.raises(RuntimeError):
        c._share_id()
    with open(path, "wb") as f:                       # this launch's, written long before this rank looks
        f.write(b"\x09" * 128 + c._launch_nonce())
    old = time.time() - 3600.0
    os.utime(path, (old, old))
    assert c._share_id() == b"\x09" * 128


def test_in_process_comm_does_not_hang_on_a_failed_rank():
    """InProcessComm: a rank that never arrives (its thread raised) ends the others' wait with an error after the
    communicator's time limit, and a rank that fails inside a collective releases the waiting ranks at once."""
    import threading
    import time
    from smcnuts_amd.parallel import InProcessComm
    g = InProcessComm(2, timeout=0.5)
    t0 = time.time()
    with pytest.raises(RuntimeError):
        g.view(0).allgather(np.ones(2))               # rank 1 never comes
    assert time.time() - t0 < 5.0
    g = InProcessComm(2, timeout=30.0)
    err = {}

    def waiter():
        try:
            g.view(0).allgather(np.ones(2))
        except RuntimeError as e:
            err["e"] = e
    th = threading.Thread(target=waiter)
    th.start()
    time.sleep(0.1)
    with pytest.raises(ZeroDivisionError):
        g.view(1)._guard(lambda: 1 / 0)               # rank 1 fails in its part of a collective
    th.join(timeout=5.0)
    assert not th.is_alive() and "e" in err


def test_exchange_sides_without_a_buffer_still_join_the_collective():
    """TorchDistComm.exchange (device path): a rank that serves no requests of the routed resampling may hand over a null
    pointer for its empty side; the side becomes an empty tensor (every rank still enters all_to_all_single), and a
    NON-empty side without a buffer is an error on that rank, not a TypeError deep inside torch."""
    import ctypes as C
    import torch
    from smcnuts_amd.parallel import TorchDistComm
    comm = TorchDistComm.__new__(TorchDistComm)
    comm._torch, comm.device = torch, torch.device("cpu")
    for ptr in (None, 0, C.c_void_p()):
        t = comm._side(ptr, 0)
        assert t.numel() == 0 and t.dtype == torch.float64
    with pytest.raises(ValueError):
        comm._side(None, 3)


def test_bench_launcher_gives_up_on_stuck_ranks(tmp_path, monkeypatch):
    """bench.py --gpus N without a launcher: ranks that never finish are killed as a process group after
    --launch-timeout and the call returns 124 with their output, instead of hanging the caller."""
    import importlib
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    stuck = tmp_path / "stuck.py"
    stuck.write_text("import time, sys\nprint('rank alive', flush=True)\ntime.sleep(600)\n")
    monkeypatch.setattr(bench.os.path, "abspath", lambda p: str(stuck) if str(p).endswith("bench.py") else os.path.realpath(p))
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    t0 = time.time()
    rc = bench.launch_ranks(2, 8.0)
    assert rc == 124 and time.time() - t0 < 60
