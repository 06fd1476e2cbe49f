"""Targets evaluated on the host: any object with the reference's StanModel
surface drives the GPU path through the density callback of the C ABI.  The "user model" here
is the oracle's C density behind that surface (tests may use the oracle); the expected values
are the golden vectors recorded from the reference."""
import os

import numpy as np
import pytest

from _tol import close

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "smcnuts_amd", "model", "data")


def host_model(name):
    """A plain Python object with .dim / .logpdf / .logpdfgrad / .constrain: no device functor."""
    if name.startswith("gauss4"):
        return orc.OracleTarget(orc.MODEL_GAUSS, orc.gauss_data(4), 4)
    if name.startswith("tgauss3"):
        return orc.OracleTarget(orc.MODEL_GAUSS, orc.gauss_data(3, prior_sd=3.0, lik_mean=1.5, lik_sd=0.5), 3)
    if name.startswith("arma"):
        return orc.OracleTarget(orc.MODEL_ARMA, orc.arma_data(os.path.join(DATA, "arma.json")), 4)
    raise KeyError(name)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


@pytest.mark.parametrize("name", ["gauss4_fwd", "tgauss3_fwd_temp", "arma_fwd", "gauss4_deep"])
def test_host_target_nuts_on_reference_tapes(golden_dir, name):
    """The tree building on the device with the density asked from the host in lock step consumes
    exactly the reference's draws and lands on its x', r' (trees of up to 2047 leapfrogs)."""
    from smcnuts_amd.proposal.nuts import NUTSProposal
    g = load(golden_dir, name)
    model = host_model(name)
    prop = NUTSProposal(model, None, float(g["eps"]))
    assert prop.target.host_evaluated and prop.target.target is model
    for k in range(min(int(g["K"]), 3)):
        xn, rn = prop.rvs(g[f"x_in_{k}"], g[f"r_{k}"], float(g[f"phi_prop_{k}"]),
                          tape=g[f"tape_{k}"], tape_off=g[f"tape_off_{k}"])
        st = prop.last_stats
        assert not st["flags"].any()
        np.testing.assert_array_equal(st["ndraws"], np.diff(g[f"tape_off_{k}"]))
        close(xn, g[f"x_new_{k}"], rtol=1e-13, atol=1e-14)
        close(rn, g[f"r_new_{k}"], rtol=2e-13, atol=2e-14)
        ref = orc.nuts_rvs(model, g[f"x_in_{k}"], g[f"r_{k}"], float(g[f"phi_prop_{k}"]), float(g["eps"]),
                           tape=g[f"tape_{k}"], tape_off=g[f"tape_off_{k}"])
        np.testing.assert_array_equal(st["nleap"], ref["nleap"])
        np.testing.assert_array_equal(st["depth"], ref["depth"])
    assert prop.target.calls >= int(st["nleap"].max()) + 1        # one callback per lock-step leapfrog


@pytest.mark.parametrize("name", ["gauss4_gaussL", "tgauss3_gaussL_temp", "arma_fwd", "arma_fwd_temp", "arma_asym_temp"])
def test_host_target_full_loop_on_reference_draws(golden_dir, name):
    """SMCSampler(target=<host model>) through all three strategies, with and without tempering:
    the population operations run on the device on the density parts the callback delivers."""
    from smcnuts_amd import SMCSampler
    g = load(golden_dir, name)
    K, N = int(g["K"]), int(g["N"])
    smc = SMCSampler(K=K, N=N, target=host_model(name), step_size=float(g["eps"]), lkernel=str(g["lkernel"]),
                     tempering=bool(g["tempering"]), x0=g["x0"], logq0=g["logq0"], seed=1)
    assert not smc.device_resident
    asym = str(g["lkernel"]) == "asymptoticLKernel"
    for k in range(K):
        u = g[f"u_resample_{k}"]
        smc.step(tape=g[f"tape_{k}"], tape_off=g[f"tape_off_{k}"], r=g[f"r_{k}"], u_resample=u if u.size else None,
                 u_accept=g[f"u_accept_{k}"] if asym else None)
        assert bool(smc.resampled[k]) == bool(g[f"resampled_{k}"])
    smc.finalise(u_final=g["u_final"] if asym else None)
    close(smc.phi, g["phi"], rtol=1e-12, atol=1e-15)
    close(smc.x_saved, g["x_saved"], rtol=1e-10, atol=1e-11)
    close(smc.logw_saved, g["logw_saved"], rtol=5e-12, atol=5e-11)
    close(smc.ess, g["ess"], rtol=1e-11)
    close(smc.mean_estimate, g["mean_estimate"], rtol=1e-11, atol=1e-13)
    close(smc.variance_estimate, g["variance_estimate"], rtol=1e-11, atol=1e-13)
    close(smc.acceptance_rate, g["acceptance_rate"], atol=1e-12)


def test_host_target_equals_device_functor_in_production_mode():
    """Philox draws: the same Gaussian once as the device functor and once as a host model."""
    from smcnuts_amd import GaussianTarget, SMCSampler
    kw = dict(K=4, N=512, step_size=0.2, seed=3)
    dev = SMCSampler(target=GaussianTarget(4), **kw)
    dev.sample(show_progress=False)
    host = SMCSampler(target=host_model("gauss4"), **kw)
    host.sample(show_progress=False)
    np.testing.assert_array_equal(dev.leapfrogs, host.leapfrogs)
    close(host.x_saved, dev.x_saved, rtol=1e-12, atol=1e-13)
    close(host.ess, dev.ess, rtol=1e-12)
    close(host.mean_estimate, dev.mean_estimate, rtol=1e-11, atol=1e-13)


def test_host_target_errors_surface():
    """An exception inside the user's model aborts the entry point with an error, not a crash."""
    from smcnuts_amd import SMCSampler

    class Broken:
        dim = 2
        def logpdf(self, x, phi=1.0):
            raise FloatingPointError("user model failed")
        def logpdfgrad(self, x, phi=1.0):
            return np.zeros_like(x)

    with pytest.raises(RuntimeError):
        SMCSampler(K=2, N=64, target=Broken(), step_size=0.1, seed=1)
    with pytest.raises(TypeError):
        SMCSampler(K=2, N=64, target=object(), step_size=0.1)


@pytest.mark.parametrize("lkernel,tempering", [("forwardsLKernel", False), ("GaussianApproxLKernel", True)])
def test_host_target_on_two_shards_equals_one_shard(lkernel, tempering):
    """A host-evaluated target with the population split over two shards (each shard calls ITS model for its own
    particles; Philox keyed by the global particle index): the run one shard of N particles makes."""
    from smcnuts_amd import SMCSampler
    from tests.test_sharding import _run_shards
    kw = dict(K=4, N=1024, step_size=0.05, seed=3, lkernel=lkernel, tempering=tempering)
    one = SMCSampler(target=host_model("arma"), **kw)
    one.sample(show_progress=False)
    assert any(one.resampled)
    sh = _run_shards(lambda c: SMCSampler(target=host_model("arma"), comm=c, **kw), 2,
                     lambda s: s.sample(show_progress=False))
    for s in sh:
        assert list(s.resampled) == list(one.resampled)
        close(s.ess, one.ess, rtol=1e-11)
        close(s.mean_estimate, one.mean_estimate, rtol=1e-10, atol=1e-13)
        close(s.variance_estimate, one.variance_estimate, rtol=1e-9, atol=1e-13)
    close(np.concatenate([s.x_saved for s in sh], axis=1), one.x_saved, rtol=1e-10, atol=1e-13)
    assert sum(int(s.leapfrogs.sum()) for s in sh) == int(one.leapfrogs.sum())
