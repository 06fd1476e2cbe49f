"""GPU parity tests proper: the HIP path, called through the C ABI, against the
oracle and the golden vectors recorded from the reference.  Tree decisions, draw
counts and resampling indices are exact.  Floating-point tolerances are the ones in
force at each call site (every one goes through tests/_tol.py::close, which records
the error observed; tools/tolerance_table.py keeps each literal <= 10x that error;
DESIGN.md 2 has the table by class): on the reference's recorded draws x' 1e-12,
r' 1e-11 (every leapfrog adds eps x the gradient's own 5e-13 relative rounding: the one
output above the 1e-12 contract, see the test), density parts 2e-12; over the K-iteration loop x_saved 1e-10
(device-resident loop 1e-12), logw / log-likelihood 1e-11 / 1e-12, ESS 1e-11, mean /
variance estimates 1e-10 / 2e-11; PRMwCD trajectories looser, for the reason given at
each test (singular prior gradient at Beta_j = 0)."""
import os

import numpy as np
import pytest

from _tol import close

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "smcnuts_amd", "model", "data")
# PRMwCD is not in this list: its trees are 500-1000 leapfrogs long through a prior
# whose gradient is singular at Beta_j = 0, i.e. chaotic -- fp64 round-off differences
# (FMA, summation order, libm) between two correct implementations grow to O(1) along
# one trajectory.  Its parity is asserted on the density/gradient and on short trees
# (test_prmwcd_*); the deep-tree logic itself on "gauss4_deep" (harmonic, non-chaotic).
CASES = ["gauss4_fwd", "gauss32_fwd", "gauss4_gaussL", "tgauss3_fwd_temp", "tgauss3_gaussL_temp", "arma_fwd",
         "gauss4_deep", "gauss256_fwd", "arma_gaussL_temp", "arma_fwd_temp", "arma_gaussL"]


def same_run_to_rounding(a, b, what=""):
    """Two runs of the same chain under DIFFERENT schedules with wide_eval on: the lane-group evaluation re-associates
    likelihood sums depending on the schedule, so a comparison that is a tie to rounding (slice, U-turn, accept) may fall
    the other way for a particle, which then differs by O(1) from there on.  On the tests' seeds this does not happen
    (the exact assertions of the wide_eval=False twins pin the control logic); the comparison is written so that it would
    not turn red if it did: at most one particle in a thousand may differ, everything else to rounding, the estimates at
    Monte-Carlo tolerance."""
    assert a.resampled == b.resampled, what
    xa, xb = np.asarray(a.x_saved), np.asarray(b.x_saved)
    bad = np.any(np.abs(xa - xb) > 1e-9 * (1.0 + np.abs(xb)), axis=(0, 2))        # per particle, any generation
    assert bad.sum() <= max(1, xa.shape[1] // 1000), f"{what}: {int(bad.sum())} of {xa.shape[1]} particles differ"
    ok = ~bad
    close(xa[:, ok], xb[:, ok], rtol=1e-9, atol=1e-10)      # (re-associated sums: rounding, not the tests' last observed value)
    assert abs(int(a.leapfrogs.sum()) - int(b.leapfrogs.sum())) <= 2048 * int(bad.sum()), what
    if not bad.any():
        np.testing.assert_array_equal(a.leapfrogs, b.leapfrogs, err_msg=what)
    close(a.ess, b.ess, rtol=1e-8 if not bad.any() else 1e-2)
    sd = np.sqrt(np.maximum(b.variance_estimate, 1e-300))
    assert np.all(np.abs(a.mean_estimate - b.mean_estimate) <= 1e-9 + (1e-2 * sd if bad.any() else 1e-8 * (1 + np.abs(b.mean_estimate)))), what



def targets(name):
    from smcnuts_amd import ArmaModel, GaussianTarget, PRMwCDModel
    if name.startswith("prmwcd"):
        return PRMwCDModel(), orc.OracleTarget(orc.MODEL_PRMWCD, orc.prmwcd_data(os.path.join(DATA, "PRMwCD.json")), 13)
    if name.startswith("gauss4"):
        return GaussianTarget(4), orc.OracleTarget(orc.MODEL_GAUSS, orc.gauss_data(4), 4)
    if name.startswith("gauss256"):
        return GaussianTarget(256), orc.OracleTarget(orc.MODEL_GAUSS, orc.gauss_data(256), 256)
    if name.startswith("gauss32"):
        return GaussianTarget(32), orc.OracleTarget(orc.MODEL_GAUSS, orc.gauss_data(32), 32)
    if name.startswith("arma_asym"):
        return ArmaModel(), orc.OracleTarget(orc.MODEL_ARMA, orc.arma_data(os.path.join(DATA, "arma.json")), 4)
    if name.startswith("tgauss3"):
        return (GaussianTarget(3, prior_sd=3.0, lik_mean=1.5, lik_sd=0.5),
                orc.OracleTarget(orc.MODEL_GAUSS, orc.gauss_data(3, prior_sd=3.0, lik_mean=1.5, lik_sd=0.5), 3))
    if name.startswith("arma"):
        return ArmaModel(), orc.OracleTarget(orc.MODEL_ARMA, orc.arma_data(os.path.join(DATA, "arma.json")), 4)
    raise KeyError(name)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


@pytest.mark.parametrize("name", ["gauss4_fwd", "gauss32_fwd", "tgauss3_fwd_temp", "arma_fwd", "prmwcd_gaussL_temp"])
@pytest.mark.parametrize("phi", [1.0, 0.37])
def test_target_value_and_gradient(name, phi):
    t, ot = targets(name)
    rng = np.random.default_rng(5)
    x = rng.normal(size=(777, t.dim)) * 0.6
    if name.startswith("arma"):
        x += np.array([0, 0.5, 0, -1.0])
        x[0] = [0.0, 0.0, 0.0, 800.0]           # sigma overflows -> -inf convention
        x[1] = [np.nan, 0.0, 0.0, 0.0]
    lp, g = t.logpdf(x, phi), t.logpdfgrad(x, phi)
    close(lp, ot.logpdf(x, phi), rtol=1e-12, atol=1e-10)
    close(g, ot.logpdfgrad(x, phi), rtol=5e-12, atol=5e-11)
    a, b = t.logpdf_parts(x[2:])
    oa, ob = ot.parts(x[2:])
    close(a, oa, rtol=1e-14, atol=1e-13)
    close(b, ob, rtol=1e-12, atol=1e-10)
    close(t.constrain(x[2:]), ot.constrain(x[2:]), rtol=1e-15)
    assert np.isscalar(t.logpdf(x[5], phi)) and t.logpdfgrad(x[5], phi).shape == (t.dim,)


@pytest.mark.parametrize("name", CASES)
def test_nuts_transition_on_reference_tapes(golden_dir, name):
    """NUTSProposal.rvs on the draws recorded from the reference: every particle
    consumes exactly the recorded number of draws (same tree, same decisions)
    and lands on the reference's x', r'."""
    from smcnuts_amd.proposal.nuts import NUTSProposal
    g = load(golden_dir, name)
    t, ot = targets(name)
    prop = NUTSProposal(t, None, float(g["eps"]))
    for k in range(int(g["K"])):
        xn, rn = prop.rvs(g[f"x_in_{k}"], g[f"r_{k}"], float(g[f"phi_prop_{k}"]),
                          tape=g[f"tape_{k}"], tape_off=g[f"tape_off_{k}"])
        st = prop.last_stats
        assert not st["flags"].any()
        np.testing.assert_array_equal(st["ndraws"], np.diff(g[f"tape_off_{k}"]))
        close(xn, g[f"x_new_{k}"], rtol=1e-12, atol=1e-13)
        # r' = r + eps/2 (g_0 + 2 sum g_i + g_end): every leapfrog adds eps x the gradient's own rounding (5e-13 relative
        # of gradients of order 1e2 for arma: FMA contraction and the chunked order of the 200-step sensitivity recurrences
        # against the oracle's plain loop), so the momentum is the one output that sits above the 1e-12 contract --
        # observed 6.2e-12 relative, 1.2e-12 absolute (arma_gaussL_temp, ~30 leapfrogs); the positions, which see that
        # error times eps again, meet it
        close(rn, g[f"r_new_{k}"], rtol=1e-11, atol=2e-12)
        ref = orc.nuts_rvs(ot, g[f"x_in_{k}"], g[f"r_{k}"], float(g[f"phi_prop_{k}"]), float(g["eps"]),
                           tape=g[f"tape_{k}"], tape_off=g[f"tape_off_{k}"])
        np.testing.assert_array_equal(st["nleap"], ref["nleap"])
        np.testing.assert_array_equal(st["depth"], ref["depth"])
        for key in ("lpri0", "llik0", "lpri1", "llik1"):
            close(st[key], ref[key], rtol=2e-12, atol=2e-11)


@pytest.mark.parametrize("name", CASES + ["tgauss3_asym_temp", "arma_asym_temp"])
def test_full_loop_on_reference_draws(golden_dir, name):
    """SMCSampler in the reference's order with the reference's draws."""
    from smcnuts_amd import SMCSampler
    g = load(golden_dir, name)
    t, _ = targets(name)
    K, N = int(g["K"]), int(g["N"])
    smc = SMCSampler(K=K, N=N, target=t, step_size=float(g["eps"]), lkernel=str(g["lkernel"]),
                     tempering=bool(g["tempering"]), x0=g["x0"], logq0=g["logq0"], seed=1)
    asym = str(g["lkernel"]) == "asymptoticLKernel"
    for k in range(K):
        u = g[f"u_resample_{k}"]
        smc.step(tape=g[f"tape_{k}"], tape_off=g[f"tape_off_{k}"], r=g[f"r_{k}"], u_resample=u if u.size else None,
                 u_accept=g[f"u_accept_{k}"] if asym else None)
        assert bool(smc.resampled[k]) == bool(g[f"resampled_{k}"])
    smc.finalise(u_final=g["u_final"] if asym else None)
    close(smc.phi, g["phi"], rtol=1e-12, atol=1e-15)
    close(smc.x_saved, g["x_saved"], rtol=1e-10, atol=1e-11)
    close(smc.logw_saved, g["logw_saved"], rtol=1e-11, atol=1e-10)
    close(smc.ess, g["ess"], rtol=1e-11)
    close(smc.log_likelihood, g["log_likelihood"], rtol=1e-12, atol=1e-12)
    close(smc.mean_estimate, g["mean_estimate"], rtol=1e-10, atol=1e-12)
    close(smc.variance_estimate, g["variance_estimate"], rtol=2e-11, atol=2e-13)
    close(smc.acceptance_rate, g["acceptance_rate"], atol=1e-12)


def test_prmwcd_poisson_edge_cases_of_the_unrolled_observation_loop():
    """poisson_lpmf's edge cases in the device functor's unrolled loop for the shipped data shape, which decides them ONCE
    behind the loop (from max mu and min(mu + [y == 0])) instead of per observation: a rate that overflows (intercept 800:
    mu = inf) and a rate that underflows to 0 where counts are not 0 (intercept -900) both give log-likelihood -inf, as the
    reference's adapter does (PRMwCD.stan:24-33 through bridgestan.py:45-49); particles next to them are untouched."""
    t, ot = targets("prmwcd_gaussL_temp")
    rng = np.random.default_rng(11)
    x = rng.normal(size=(64, t.dim)) * 0.3
    x[5, 0] = 800.0
    x[9, 0] = -900.0
    x[12, 0] = 690.0            # large but finite: mu ~ 1e300
    for phi in (1.0, 0.4):
        lp, olp = t.logpdf(x, phi), ot.logpdf(x, phi)
        assert lp[5] == -np.inf and lp[9] == -np.inf and olp[5] == -np.inf and olp[9] == -np.inf
        ok = np.isfinite(olp)
        assert ok.sum() >= 61
        close(lp[ok], olp[ok], rtol=1e-13, atol=1e-12)
        a, b = t.logpdf_parts(x)
        oa, ob = ot.parts(x)
        assert b[5] == -np.inf and b[9] == -np.inf and np.all(np.isfinite(a))
        close(b[ok], ob[ok], rtol=1e-13, atol=1e-12)


@pytest.mark.parametrize("cap,widen,requeue", [(0, 0, 0), (2, 0, 0), (1, 1, 0), (3, 1, 0), (0, 2, 0), (3, 1, 2), (3, 0, 1)])
def test_prmwcd_short_trees_match_oracle(golden_dir, cap, widen, requeue):
    """PRMwCD (BASELINE config 4 target): NUTS with max_depth 4 (<= 31 leapfrogs) from
    the particle states the reference visited; Philox on both sides; decisions exact.
    cap > 0 (smcn_set_nuts_cap): trees that want more than `cap` doublings are parked at that boundary and finished by a
    second launch -- widen = 0 by the same 8-lanes-per-particle kernel, widen = 1 by the wavefront-per-particle functor
    (100 observations over 64 lanes): the same trees and, at these lengths, the same states either way.  widen = 2:
    nothing is parked, every tree runs from its start in the finisher's kernel."""
    from smcnuts_amd import _capi
    import ctypes as C
    g = load(golden_dir, "prmwcd_gaussL_temp")
    t, ot = targets("prmwcd_gaussL_temp")
    x = np.concatenate([g["x_saved"][k] for k in range(int(g["K"]) + 1)])
    x = np.tile(x, (8, 1))
    N = x.shape[0]
    ctx = _capi.Context(N, t.model_id, t.model_data)
    ctx.set_seed(77)
    ctx.set_state(x=x, logw=np.zeros(N))
    ctx.call("smcn_set_nuts_cap", cap, widen)
    ctx.call("smcn_set_nuts_requeue", requeue)    # (an inner park level of the first launch: the same trees)
    for phi, it in ((1.0, 0), (0.13, 1)):
        ctx.propose_nuts(0.01, phi, it, max_depth=4)
        parked = C.c_int64(-1)
        ctx.call("smcn_nuts_parked", C.byref(parked))
        assert (parked.value > N // 10) if cap else (parked.value == 0)
        r, xn, rn, _ = ctx.get_proposal()
        st = ctx.tree_stats()
        ref = orc.nuts_rvs(ot, x, r, phi, 0.01, max_depth=4, seed=77, iteration=it)
        np.testing.assert_array_equal(st["ndraws"], ref["ndraws"])
        np.testing.assert_array_equal(st["nleap"], ref["nleap"])
        # the prior gradient ~ |Beta_j|^(-1/2) amplifies round-off for coordinates passing near 0:
        # 1e-6 here (observed worst 1.5e-8 abs), against 1e-9/1e-10 for the smooth targets
        close(xn, ref["x_new"], rtol=1e-6, atol=1e-7)
        close(rn, ref["r_new"], rtol=1e-6, atol=1e-6)
        assert np.mean(np.abs(xn - ref["x_new"]) < 1e-10) > 0.999
        lp0, ll0, lp1, ll1 = ctx.density_parts()
        close(lp1, ref["lpri1"], rtol=1e-7, atol=1e-7)
        close(ll1, ref["llik1"], rtol=5e-9, atol=5e-9)


@pytest.mark.parametrize("cap,widen,requeue", [(0, 0, 0), (9, 0, 0), (9, 1, 0), (6, 1, 0), (4, 1, 0), (0, 2, 0),
                                               (9, 1, 8), (9, 1, 3), (9, 0, 6), (6, 1, 5)])
def test_prmwcd_deep_trees_at_a_small_step_match_oracle(golden_dir, cap, widen, requeue):
    """The PRMwCD functors through EVERY level of their tree stacks (LDS levels, then the HBM slots), against the oracle: chaos
    grows with the integration TIME, not with the number of leapfrogs, so at a step of 1e-4 the full-depth trees
    (11 doublings, 2 047 leapfrogs, total time 0.2 -- the horizon of the 31-leapfrog trees at the production step) stay on the
    oracle's decisions: draws consumed (= every merge, slice test and accept), leapfrog counts and depths exact, the
    selected states to 1e-9.  Production trees (step 0.01) average 325 leapfrogs and diverge from ANY second
    implementation after ~50; this is the same code path at a horizon where parity is decidable.
    (cap, widen) = smcn_set_nuts_cap: (0, 0) the one-launch 8-lane kernel; (9, 1) what BASELINE config 4 SHIPS
    (PRMwCDModel.two_phase_default): trees that want more than 9 doublings are parked and the 2 047-leapfrog ones are
    finished by nuts_kernel<PrmwcdDistModel<64, ..., 5, FAST>, hybrid, TWO_PHASE> -- resume from the parked record, eval_wave,
    its 5 LDS stack levels and the HBM levels above them; (6, 1) / (4, 1) park earlier, so that the finisher also builds the
    middle doublings; (9, 0) finishes with the kernel that parked; (0, 2) runs every tree from its start in the finisher's
    kernel.  If the finisher's trees left the oracle's at this horizon where the 8-lane kernel's do not, that would be a
    bug, not chaos.  requeue > 0 (smcn_set_nuts_requeue; (9, 1, 8) is what config 4 ships since round 5): the first launch
    parks its trees at that inner level as well and its own groups take them up again -- record written through by one
    compute unit, read by another -- before they reach the cap."""
    from smcnuts_amd import _capi
    import ctypes as C
    g = load(golden_dir, "prmwcd_gaussL_temp")
    t, ot = targets("prmwcd_gaussL_temp")
    x = np.concatenate([g["x_saved"][k] for k in range(int(g["K"]) + 1)])[:192]
    N = x.shape[0]
    ctx = _capi.Context(N, t.model_id, t.model_data)
    ctx.set_seed(404)
    ctx.set_state(x=x, logw=np.zeros(N))
    ctx.call("smcn_set_nuts_cap", cap, widen)
    ctx.call("smcn_set_nuts_requeue", requeue)
    for phi, it, eps in ((1.0, 0, 1e-4), (0.2, 1, 1e-4)):
        ctx.propose_nuts(eps, phi, it)
        parked = C.c_int64(-1)
        ctx.call("smcn_nuts_parked", C.byref(parked))
        assert (parked.value > N // 2) if cap else (parked.value == 0)     # (most trees here want all 11 doublings)
        r, xn, rn, _ = ctx.get_proposal()
        st = ctx.tree_stats()
        ref = orc.nuts_rvs(ot, x, r, phi, eps, seed=404, iteration=it)
        mism = np.flatnonzero(st["ndraws"] != ref["ndraws"])
        assert mism.size == 0, f"particles {mism.tolist()} took another tree: {st['ndraws'][mism].tolist()} vs {ref['ndraws'][mism].tolist()}"
        np.testing.assert_array_equal(st["nleap"], ref["nleap"])
        np.testing.assert_array_equal(st["depth"], ref["depth"])
        assert (st["nleap"] >= 1023).sum() > N // 2          # the HBM levels of the stack were exercised
        assert phi < 1.0 or (st["depth"] <= 9).sum() > 10    # ... and sub-tree U-turns ended other trees early
        # a particle or two pass close to Beta_j = 0 (singular prior gradient) and amplify the last bits even over this
        # horizon (observed: 1e-7 on one particle, everything else < 1e-11): the bulk is pinned tightly, the rest loosely
        assert np.mean(np.abs(xn - ref["x_new"]).max(axis=1) < 1e-10) > 0.98
        close(xn, ref["x_new"], rtol=5e-7, atol=5e-8)
        close(rn, ref["r_new"], rtol=5e-6, atol=5e-6)
        lp0, ll0, lp1, ll1 = ctx.density_parts()
        close(lp1, ref["lpri1"], rtol=2e-7, atol=2e-6)
        close(ll1, ref["llik1"], rtol=1e-9, atol=1e-8)


@pytest.mark.parametrize("widen", [0, 2])
def test_prmwcd_teacher_forced_leapfrogs_along_long_trajectories(golden_dir, widen):
    """Config 4's production trees, teacher-forced: the oracle integrates 8 posterior particles for 700 leapfrogs at the
    production step (0.01) in each direction -- the states a 1 000+-leapfrog tree visits, beta coordinates crossing 0 where
    the prior's gradient is singular included -- and the device advances ONE leapfrog from each of those 11 208 states
    (a tree of depth 0 on a recorded tape: the slice admits the leaf, the accept draw is 0).  No chaos over one step:
    positions, momenta and both density parts to 1e-12.  widen = 0: the 8-lanes-per-particle kernel; widen = 2
    (smcn_set_nuts_cap): the kernel that finishes config 4's parked trees -- a wavefront per particle, eval_wave."""
    from smcnuts_amd import _capi
    g = load(golden_dir, "prmwcd_gaussL_temp")
    t, ot = targets("prmwcd_gaussL_temp")
    x0 = g["x_saved"][int(g["K"])][:8]
    r0 = np.random.default_rng(8).standard_normal(x0.shape)
    eps, T = 0.01, 700

    def tapes(n, u_dir):
        return np.tile([60.0, u_dir, 0.0], n), 3 * np.arange(n + 1, dtype=np.int64)

    xs, rs, nx, nr, nlp, nll, dirs = [], [], [], [], [], [], []
    for u_dir in (0.25, 0.75):                      # forward, backward (nuts.py:91: direction = +1 iff u < 0.5)
        x, r = x0.copy(), r0.copy()
        tape, off = tapes(len(x), u_dir)
        for _ in range(T):
            ref = orc.nuts_rvs(ot, x, r, 1.0, eps, max_depth=0, tape=tape, tape_off=off)
            assert np.all(ref["nleap"] == 1)
            moved = np.any(ref["x_new"] != x, axis=1)
            assert moved.all()                      # the leaf was admitted by the slice and accepted
            xs.append(x); rs.append(r); nx.append(ref["x_new"]); nr.append(ref["r_new"])
            nlp.append(ref["lpri1"]); nll.append(ref["llik1"]); dirs.append(np.full(len(x), u_dir))
            x, r = ref["x_new"], ref["r_new"]
    X, R, NX, NR = (np.concatenate(a) for a in (xs, rs, nx, nr))
    NLP, NLL, U = np.concatenate(nlp), np.concatenate(nll), np.concatenate(dirs)
    assert np.abs(nx[T - 1] - x0).max() > 0.3           # the orbits travel
    M = X.shape[0]
    ctx = _capi.Context(M, t.model_id, t.model_data)
    ctx.set_state(x=X, logw=np.zeros(M))
    ctx.call("smcn_set_nuts_cap", 0, widen)
    ctx.call("smcn_set_momentum", _capi.dptr(np.ascontiguousarray(R)))
    tape = np.stack([np.full(M, 60.0), U, np.zeros(M)], axis=1).reshape(-1)
    ctx.propose_nuts(eps, 1.0, 0, max_depth=0, tape=tape, tape_off=3 * np.arange(M + 1, dtype=np.int64))
    _, xn, rn, _ = ctx.get_proposal(r=False)
    assert np.all(ctx.tree_stats()["nleap"] == 1)
    close(xn, NX, rtol=1e-14, atol=5e-16)
    close(rn, NR, rtol=5e-13, atol=5e-14)
    _, _, lp1, ll1 = ctx.density_parts()
    close(lp1, NLP, rtol=5e-14, atol=5e-14)
    close(ll1, NLL, rtol=2e-14, atol=2e-13)


def test_two_phase_launch_equals_one_launch():
    """smcn_set_nuts_cap(doublings, widen = 0): parking the long trees of a PRMwCD launch at a doubling boundary and finishing
    them with the same kernel in a second launch changes NOTHING -- every output of a whole tempered run with the Gaussian
    L-kernel (trees up to 2 047 leapfrogs, chaotic trajectories) is bit-identical to the one-launch run."""
    from smcnuts_amd import PRMwCDModel, SMCSampler
    import ctypes as C

    def run(cap):
        smc = SMCSampler(K=5, N=4096, target=PRMwCDModel(), step_size=0.01, lkernel="GaussianApproxLKernel", tempering=True,
                         seed=10)
        smc.samples.ctx.call("smcn_set_nuts_cap", cap, 0)
        parked = []
        for _ in range(5):
            smc.step()
            v = C.c_int64(0)
            smc.samples.ctx.call("smcn_nuts_parked", C.byref(v))
            parked.append(v.value)
        smc.finalise()
        return smc, parked

    one, p0 = run(0)
    for cap in (9, 6):
        two, p = run(cap)
        assert p0 == [0] * 5 and min(p) > 0
        for name in ("x_saved", "logw_saved", "ess", "phi", "mean_estimate", "variance_estimate", "leapfrogs", "log_likelihood"):
            np.testing.assert_array_equal(getattr(one, name), getattr(two, name), err_msg=f"cap {cap}: {name}")


def test_inner_park_level_gives_the_same_run():
    """smcn_set_nuts_requeue: trees parked once more INSIDE the first launch and taken up again by its own groups -- the record
    written through by one compute unit and read by another, 20 000+ hand-overs a launch at this size -- are the same trees:
    every output of a tempered config-4 run (wave-per-tree finisher as shipped, 40 000 particles = more than twice the
    16 384 resident lane groups, so that the queue of parked trees fills while fresh particles are still handed out) is
    bit-identical with the level off, at 8 doublings (shipped) and at 5."""
    from smcnuts_amd import PRMwCDModel, SMCSampler
    import ctypes as C

    def run(requeue, N, K):
        smc = SMCSampler(K=K, N=N, target=PRMwCDModel(), step_size=0.01, lkernel="GaussianApproxLKernel", tempering=True,
                         seed=10, nuts_cap=(9, True, requeue))
        parked = []
        for _ in range(K):
            smc.step()
            v = C.c_int64(0)
            smc.samples.ctx.call("smcn_nuts_parked", C.byref(v))
            parked.append(v.value)
        smc.finalise()
        return smc, parked

    # (2 048 particles, 8 iterations: the records of ALL hand-overs -- 2 MB -- fit the L2 of one XCD, and every particle's slot is
    #  rewritten in every launch: a record served from a cache line of the launch before would show here)
    for N, K in ((40000, 4), (2048, 8)):
        off, p0 = run(0, N, K)
        assert min(p0) > 0
        for requeue in (8, 5):
            on, p = run(requeue, N, K)
            assert p == p0
            for name in ("x_saved", "logw_saved", "ess", "phi", "mean_estimate", "variance_estimate", "leapfrogs", "log_likelihood"):
                np.testing.assert_array_equal(getattr(off, name), getattr(on, name), err_msg=f"N {N}, requeue {requeue}: {name}")
    assert PRMwCDModel().two_phase_default == (9, True, 8)


def test_prmwcd_config4_runs_to_phi_one():
    """BASELINE config 4 shape (PRMwCD, Gaussian L-kernel + adaptive tempering),
    reduced N: the temperature ladder is monotone and reaches 1, outputs finite."""
    from smcnuts_amd import PRMwCDModel, SMCSampler
    smc = SMCSampler(K=14, N=4096, target=PRMwCDModel(), step_size=0.01, lkernel="GaussianApproxLKernel",
                     tempering=True, seed=5, save_history=False)
    smc.sample(show_progress=False)
    assert np.all(np.diff(smc.phi) >= 0) and smc.phi[0] > 0 and smc.phi[-1] == 1.0
    assert np.all(np.isfinite(smc.mean_estimate)) and np.all(np.isfinite(smc.log_likelihood))
    assert np.all(smc.ess >= 1.0 - 1e-9)    # (the reference degenerates the same way here: golden ess ~ 2.5 of 32)


def test_config4_weight_path_on_the_reference_proposals(golden_dir):
    """BASELINE config 4 (PRMwCD, Gaussian L-kernel + adaptive tempering) WITHOUT its chaotic trajectories:
    every iteration takes the reference's recorded proposal (r, x', r'; smcn_set_proposal) and runs the whole
    D = 13 weight path on the device -- density parts at x and x', moments of [-r', x'] + conditional Gaussian
    (gaussian_lkernel.py:24-84), ESS bisection (adaptive_tempering.py:18-63), re-weight at phi = 1
    (samples.py:183-196), normalise, ESS, multinomial ancestors, constrained estimates -- against the
    reference's recorded outputs."""
    from smcnuts_amd import SMCSampler, _capi
    g = load(golden_dir, "prmwcd_gaussL_temp")
    t, _ = targets("prmwcd_gaussL_temp")
    K, N = int(g["K"]), int(g["N"])
    smc = SMCSampler(K=K, N=N, target=t, step_size=float(g["eps"]), lkernel="GaussianApproxLKernel", tempering=True,
                     x0=g["x0"], logq0=g["logq0"], seed=1)
    s = smc.samples
    close(s.phi_new, g["phi"][0], rtol=1e-12)

    def recorded_proposal(ctx, phi, iteration, tape=None, tape_off=None, r=None):
        k = smc.k
        close(phi, float(g[f"phi_prop_{k}"]), rtol=1e-12)
        np.testing.assert_array_equal(s.x, g[f"x_in_{k}"])         # the resampled state is the reference's, bit for bit
        ctx.call("smcn_set_proposal", *(_capi.dptr(np.ascontiguousarray(g[f"{n}_{k}"])) for n in ("r", "x_new", "r_new")))

    s.forward_kernel.propose = recorded_proposal
    for k in range(K):
        u = g[f"u_resample_{k}"]
        smc.step(u_resample=u if u.size else None)
        assert bool(smc.resampled[k]) == bool(g[f"resampled_{k}"])
        close(smc.logw_saved[k + 1], g["logw_saved"][k + 1], rtol=1e-12, atol=1e-11)
    smc.finalise()
    close(smc.phi, g["phi"], rtol=1e-12, atol=1e-15)
    np.testing.assert_array_equal(smc.x_saved, g["x_saved"])
    close(smc.logw_saved, g["logw_saved"], rtol=1e-12, atol=1e-11)
    close(smc.ess, g["ess"], rtol=1e-11)
    close(smc.log_likelihood, g["log_likelihood"], rtol=1e-12, atol=1e-12)
    close(smc.mean_estimate, g["mean_estimate"], rtol=1e-11, atol=1e-13)
    close(smc.variance_estimate, g["variance_estimate"], rtol=1e-11, atol=1e-13)
    assert sum(smc.resampled) >= 3 and 0 < smc.phi[0] < smc.phi[-1] <= 1.0
    assert s.lkernel.last_path == "device"       # the D x D algebra ran on the GPU (smcn_glk.hpp), not in NumPy


def test_config4_full_size_properties_and_posterior_means():
    """BASELINE config 4 at its specified size (PRMwCD, N = 65 536, Gaussian L-kernel + adaptive tempering):
    run-to-run determinism, a monotone temperature ladder that reaches 1, finite outputs, and final estimates
    near the long-run Stan values shipped with the reference (stan_models/PRMwCD/PRMwCD.params:1-13)."""
    from smcnuts_amd import PRMwCDModel, SMCSampler
    truth = np.loadtxt(os.path.join(DATA, "PRMwCD.params"), usecols=(1, 2))
    outs = []
    for _ in range(2):
        smc = SMCSampler(K=16, N=65536, target=PRMwCDModel(), step_size=0.01, lkernel="GaussianApproxLKernel",
                         tempering=True, seed=21, save_history=False)
        smc.sample(show_progress=False)
        outs.append((smc.phi.copy(), smc.ess.copy(), smc.mean_estimate.copy(), smc.leapfrogs.copy()))
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)
    phi, ess, mean, leaps = outs[0]
    assert np.all(np.diff(phi) >= 0) and 0 < phi[0] < 1 and phi[-1] == 1.0
    assert np.all(np.isfinite(mean)) and np.all(np.isfinite(smc.log_likelihood)) and np.all(ess >= 1.0 - 1e-9)
    assert leaps.min() > 65536 * 20
    # posterior means within a QUARTER of a posterior standard deviation of the long-run Stan values (col 3 of the file is
    # the sd: the variance estimates below reproduce its square): the run ends with ESS ~ 17 000, i.e. a Monte-Carlo
    # error of ~0.01 sd (measured on this seed: max |z| 0.04, variance ratios 0.91-1.00; other seeds and K up to 32:
    # |z| <= 0.09, ratios 0.80-1.06) -- a wrong sign or constant in one prior or Jacobian term moves a mean by far more
    sd = np.maximum(truth[:, 1], 1e-3)
    z = (mean[-1] - truth[:, 0]) / sd
    assert np.all(np.abs(z) < 0.25), (z, mean[-1], truth[:, 0])
    ratio = smc.variance_estimate[-1] / sd ** 2
    assert np.all((ratio > 0.7) & (ratio < 1.3)), ratio


def test_prmwcd_forward_lkernel_through_sample():
    """SMCSampler(target=PRMwCDModel(), lkernel="forwardsLKernel").sample(): the device-resident driver
    asks the library (smcn_fused_transitions) whether the model's kernel takes several iterations per
    launch instead of assuming it; the result equals the step-by-step loop (same Philox keys)."""
    from smcnuts_amd import PRMwCDModel, SMCSampler
    kw = dict(K=3, N=1024, step_size=0.01, lkernel="forwardsLKernel", seed=4)
    a = SMCSampler(target=PRMwCDModel(), **kw)
    a.sample(show_progress=False)
    b = SMCSampler(target=PRMwCDModel(), **kw)
    for _ in range(3):
        b.step()
    b.finalise()
    assert a.resampled == b.resampled
    np.testing.assert_array_equal(a.x_saved, b.x_saved)
    np.testing.assert_array_equal(a.leapfrogs, b.leapfrogs)
    close(a.ess, b.ess, rtol=1e-13)
    close(a.mean_estimate, b.mean_estimate, rtol=2e-13, atol=2e-15)
    assert np.all(np.isfinite(a.mean_estimate)) and a.leapfrogs.min() > 1024


@pytest.mark.parametrize("name", ["gauss4_gaussL", "tgauss3_fwd_temp", "arma_fwd"])
def test_resampling_indices_bit_exact(golden_dir, name):
    """Multinomial ancestor indices: exact against the reference's recorded
    choice() (sequential cumsum) and against the oracle's blocked order."""
    from smcnuts_amd import _capi
    g = load(golden_dir, name)
    t, _ = targets(name)
    N = int(g["N"])
    ctx = _capi.Context(N, t.model_id, t.model_data)
    hit = 0
    for k in range(int(g["K"])):
        if not bool(g[f"resampled_{k}"]):
            continue
        hit += 1
        ctx.set_state(x=g["x_saved"][k], logw=g[f"logw_pre_{k}"])
        ll = np.empty(1); ess = np.empty(1)
        ctx.call("smcn_normalise", _capi.dptr(ll), _capi.dptr(ess))
        close(ll[0], g["log_likelihood"][k], rtol=1e-14)
        close(ess[0], g["ess"][k], rtol=1e-13)
        wn = ctx.get_state(x=False, logw=False, wn=True)[2]
        close(wn, g[f"wn_{k}"], rtol=1e-14)
        idx = ctx.resample(ll[0], np.log(N), k, u=g[f"u_resample_{k}"], want_idx=True)
        np.testing.assert_array_equal(idx, g[f"idx_{k}"])
        np.testing.assert_array_equal(idx, orc.multinomial_indices(wn, g[f"u_resample_{k}"], "blocked"))
        x, logw, _ = ctx.get_state()
        np.testing.assert_array_equal(x, g[f"x_in_{k}"])
        close(logw, ll[0] - np.log(N), rtol=1e-15)
    assert hit > 0


@pytest.mark.parametrize("N,D", [(5000, 16), (4097, 70), (16384, 256)])
def test_resampling_of_wide_particles_row_gather(N, D):
    """D >= 16 takes the two-stage resampling (search, then rows gathered one at a time and dealt to the XCDs):
    the ancestors are the oracle's, exactly, and every coordinate of every particle is its ancestor's --
    recorded uniforms (incl. one key beyond the last cdf value) and Philox, ragged N, D not a multiple of 8."""
    from smcnuts_amd import _capi, GaussianTarget
    t = GaussianTarget(D)
    rng = np.random.default_rng(N + D)
    ctx = _capi.Context(N, t.model_id, t.model_data)
    ctx.set_seed(5)
    x = rng.standard_normal((N, D))
    logw = 2.0 * rng.standard_normal(N)
    ll = np.empty(1); ess = np.empty(1)
    u = rng.random(N)
    u[7] = 1.0 - 2.0 ** -53                       # the largest double below 1: may land past cdf[-1] / cdf[-1]
    for tape in (u, None):
        ctx.set_state(x=x, logw=logw)
        ctx.call("smcn_normalise", _capi.dptr(ll), _capi.dptr(ess))
        wn = ctx.get_state(x=False, logw=False, wn=True)[2]
        idx = ctx.resample(ll[0], np.log(N), 2, u=tape, want_idx=True)
        if tape is not None:
            np.testing.assert_array_equal(idx, orc.multinomial_indices(wn, tape, "blocked"))
        xr, lw, _ = ctx.get_state()
        np.testing.assert_array_equal(xr, x[np.minimum(idx, N - 1)])
        close(lw, ll[0] - np.log(N), rtol=1e-15)
        assert len(np.unique(idx)) < N


@pytest.mark.parametrize("N", [1000, 65536])
def test_systematic_resampling_option(N):
    """resampling="systematic" (an option of the build, not in the reference): one uniform per
    resampling, keys (i + u0)/N on the same prefix sum and search.  Indices exact against the
    oracle; size-independent properties: ancestors sorted, every count within 1 of N*w."""
    from smcnuts_amd import _capi, GaussianTarget
    t = GaussianTarget(4)
    rng = np.random.default_rng(N)
    ctx = _capi.Context(N, t.model_id, t.model_data)
    ctx.call("smcn_set_resample_scheme", 1)
    ctx.set_seed(77)
    x = rng.standard_normal((N, 4))
    logw = 3.0 * rng.standard_normal(N)
    ctx.set_state(x=x, logw=logw)
    ll = np.empty(1); ess = np.empty(1)
    ctx.call("smcn_normalise", _capi.dptr(ll), _capi.dptr(ess))
    wn = ctx.get_state(x=False, logw=False, wn=True)[2]
    idx = ctx.resample(ll[0], np.log(N), 3, want_idx=True)
    u0 = orc.philox_particle_uniforms(77, 3, 0, 1, 2, 0)[0]          # stream 2 = resampling, particle slot 0, draw 0
    np.testing.assert_array_equal(idx, orc.systematic_indices(wn, u0, "blocked"))
    assert np.all(np.diff(idx) >= 0)
    counts = np.bincount(idx, minlength=N)
    assert np.all(np.abs(counts - N * wn) < 1.0 + 1e-6)
    xr = ctx.get_state()[0]
    np.testing.assert_array_equal(xr, x[idx])
    # recorded-uniform form: u[0] is the one draw
    ctx.set_state(x=x, logw=logw)
    ctx.call("smcn_normalise", _capi.dptr(ll), _capi.dptr(ess))
    idx2 = ctx.resample(ll[0], np.log(N), 3, u=np.full(N, 0.25), want_idx=True)
    np.testing.assert_array_equal(idx2, orc.systematic_indices(wn, 0.25, "blocked"))


def test_philox_streams_bit_exact_and_momenta():
    """Device Philox == oracle Philox: resampling uniforms enter only through
    the indices (exact); Box-Muller momenta to 1e-14."""
    from smcnuts_amd import GaussianTarget, _capi
    N, D, seed = 4096, 5, 987654321012345
    t = GaussianTarget(D)
    ctx = _capi.Context(N, t.model_id, t.model_data, particle_base=1000)
    ctx.set_seed(seed)
    rng = np.random.default_rng(3)
    logw = rng.normal(size=N) * 3
    ctx.set_state(x=rng.normal(size=(N, D)), logw=logw)
    ll = np.empty(1); ess = np.empty(1)
    ctx.call("smcn_normalise", _capi.dptr(ll), _capi.dptr(ess))
    wn = ctx.get_state(x=False, logw=False, wn=True)[2]
    idx = ctx.resample(ll[0], np.log(N), 7, want_idx=True)
    u = orc.philox_particle_uniforms(seed, 7, 1000, N, 2, 0)
    np.testing.assert_array_equal(idx, orc.multinomial_indices(wn, u, "blocked"))
    ctx.propose_nuts(0.1, 1.0, 3)
    r = ctx.get_proposal(x_new=False, r_new=False)[0]
    close(r, orc.philox_normals(seed, 3, N, D, 1, particle_base=1000), rtol=1e-13, atol=1e-14)


def test_momenta_of_wide_targets_match_the_oracle():
    """The momentum draw of the targets whose particle fills a wavefront (normals_pm_kernel: particle-major, Box-Muller from
    the lean log1p / rsqrt / sincos of smcn_device.hpp) against the oracle's Box-Muller on the same Philox keys, odd and even
    dimensions: 1e-13 as for the small targets, sign and quadrant of every pair included."""
    from smcnuts_amd import GaussianTarget, _capi
    for D, seed in ((101, 555), (256, 99)):
        N = 1500
        t = GaussianTarget(D)
        ctx = _capi.Context(N, t.model_id, t.model_data, particle_base=70)
        ctx.set_seed(seed)
        ctx.set_state(x=np.zeros((N, D)), logw=np.zeros(N))
        ctx.propose_nuts(0.1, 1.0, 6, max_depth=1)
        r = ctx.get_proposal(x_new=False, r_new=False)[0]
        close(r, orc.philox_normals(seed, 6, N, D, 1, particle_base=70), rtol=1e-13, atol=1e-14)


@pytest.mark.parametrize("model", ["arma", "gauss"])
def test_philox_mode_nuts_matches_oracle(model):
    """Production RNG: GPU and oracle run Philox on the same keys; decisions
    exact, positions to round-off.  N large enough to fill the queue logic."""
    from smcnuts_amd import ArmaModel, GaussianTarget, _capi
    N, seed = 20000, 4242
    if model == "arma":
        t, ot, eps = ArmaModel(), orc.OracleTarget(orc.MODEL_ARMA, orc.arma_data(os.path.join(DATA, "arma.json")), 4), 0.01
        x = np.random.default_rng(1).normal(size=(N, 4)) * np.array([0.05, 0.05, 0.1, 0.1]) + np.array([0, 0.9, 0, -1.8])
    else:
        t, ot, eps = GaussianTarget(7), orc.OracleTarget(orc.MODEL_GAUSS, orc.gauss_data(7), 7), 0.2
        x = np.random.default_rng(1).normal(size=(N, 7))
    ctx = _capi.Context(N, t.model_id, t.model_data)
    ctx.set_seed(seed)
    ctx.set_state(x=x, logw=np.zeros(N))
    ctx.propose_nuts(eps, 1.0, 11)
    r, xn, rn, _ = ctx.get_proposal()
    st = ctx.tree_stats()
    ref = orc.nuts_rvs(ot, x, r, 1.0, eps, seed=seed, iteration=11)
    mism = np.flatnonzero(st["ndraws"] != ref["ndraws"])
    # A different tree can only come from an ulp-level tie in a slice or U-turn comparison (device exp/log1p vs libm).
    # On these fixed seeds there is none today: any mismatch is a regression and names its particles.
    assert mism.size == 0, f"particles {mism.tolist()} took a different tree (ndraws {st['ndraws'][mism].tolist()} vs {ref['ndraws'][mism].tolist()})"
    ok = np.setdiff1d(np.arange(N), mism)
    np.testing.assert_array_equal(st["nleap"][ok], ref["nleap"][ok])
    close(xn[ok], ref["x_new"][ok], rtol=1e-12, atol=1e-13)
    close(rn[ok], ref["r_new"][ok], rtol=1e-12, atol=1e-13)
    assert ctx.last_leapfrogs() == int(st["nleap"].sum())


@pytest.mark.parametrize("nobs,C,q", [(37, 5, 0.5), (100, 3, 0.5), (8, 11, 0.5), (64, 7, 0.8), (1, 1, 0.5)])
def test_prmwcd_other_data_shapes_vs_oracle(tmp_path, nobs, C, q):
    """PRMwCD with other data than the shipped file (100 observations, 11 kernel columns are the device functor's
    CAPACITY, not its shape): density, both gradients and short NUTS trees against the oracle; data beyond the capacity
    is refused at creation with the message that names the host-evaluated route."""
    import json
    from smcnuts_amd import PRMwCDModel, _capi
    rng = np.random.default_rng(100 * nobs + C)
    X = np.exp(-rng.random((nobs, C)) * 3.0)
    y = rng.poisson(3.0, size=nobs)
    path = str(tmp_path / "prm.json")
    json.dump({"N": nobs, "M": C + 1, "Clength": C, "q": q, "y": y.tolist(), "Xkernel": X.reshape(-1).tolist()}, open(path, "w"))
    t = PRMwCDModel(path)
    D = C + 2
    assert t.dim == D
    ot = orc.OracleTarget(orc.MODEL_PRMWCD, orc.prmwcd_data(path), D)
    x = rng.normal(size=(513, D)) * 0.4
    x[3, -1] = 40.0                                   # Gamma = e^40: the prior's exp(-g) underflows gracefully
    for phi in (1.0, 0.3):
        close(t.logpdf(x, phi), ot.logpdf(x, phi), rtol=1e-14, atol=5e-12)
        close(t.logpdfgrad(x, phi), ot.logpdfgrad(x, phi), rtol=1e-12, atol=1e-11)
    N, seed, eps = 2048, 77, 0.002                    # short trajectories: PRMwCD is chaotic beyond a few dozen leapfrogs
    xs = rng.normal(size=(N, D)) * 0.3
    ctx = _capi.Context(N, t.model_id, t.model_data)
    ctx.set_seed(seed)
    ctx.set_state(x=xs, logw=np.zeros(N))
    ctx.propose_nuts(eps, 1.0, 4, max_depth=3)
    r, xn, rn, _ = ctx.get_proposal()
    st = ctx.tree_stats()
    ref = orc.nuts_rvs(ot, xs, r, 1.0, eps, seed=seed, iteration=4, max_depth=3)
    # every particle on the oracle's tree (a different one could only come from an ulp-level tie in a slice or U-turn
    # comparison; on these seeds there is none: any mismatch is a regression and names its particles)
    mism = np.flatnonzero(st["ndraws"] != ref["ndraws"])
    assert mism.size == 0, f"particles {mism.tolist()} took a different tree (ndraws {st['ndraws'][mism].tolist()} vs {ref['ndraws'][mism].tolist()})"
    np.testing.assert_array_equal(st["nleap"], ref["nleap"])
    close(xn, ref["x_new"], rtol=1e-11, atol=1e-12)
    # the same trees in two launches: parked after one doubling and finished by nuts_fin_kernel's GENERIC instantiation
    # (these shapes are not the unrolled one: the wavefront-per-tree kernel runs the functor's plain 64-lane evaluation)
    import ctypes as CT
    ctx.call("smcn_set_nuts_cap", 1, 1)
    ctx.propose_nuts(eps, 1.0, 4, max_depth=3)
    parked = CT.c_int64(-1)
    ctx.call("smcn_nuts_parked", CT.byref(parked))
    assert parked.value > N // 4
    _, xn2, rn2, _ = ctx.get_proposal()
    st2 = ctx.tree_stats()
    np.testing.assert_array_equal(st2["ndraws"], ref["ndraws"])
    np.testing.assert_array_equal(st2["nleap"], ref["nleap"])
    close(xn2, ref["x_new"], rtol=1e-11, atol=1e-12)
    close(rn2, ref["r_new"], rtol=1e-10, atol=1e-11)
    json.dump({"N": 101, "M": 3, "Clength": 2, "q": 0.5, "y": [1] * 101, "Xkernel": [0.5] * 202}, open(path, "w"))
    with pytest.raises(Exception, match="host-evaluated"):
        big = PRMwCDModel(path)
        _capi.Context(16, big.model_id, big.model_data)


def _closest_comparison(ot, x0, r0, eps, seed, iteration, particle):
    """The oracle's tree for one particle (Philox draws of that particle), re-built by the Python restatement with
    every comparison's margin recorded: returns (smallest relative margin, which comparison, leapfrog count there).
    slice: logu < joint (nuts.py:124); divergence: logu - 100 >= joint (:125); U-turn: (x+ - x-) . r < 0 (:152-160)."""
    from oracle.pynuts import PyNUTS

    class PhiloxRNG:
        def __init__(self):
            self.u = orc.philox_uniforms(seed, iteration, particle, 0, 0, 4200)
            self.q = 0
        def exponential(self, scale=1.0):
            self.q = 1
            return -np.log1p(-self.u[0])
        def uniform(self, lo=0.0, hi=1.0):
            v = self.u[self.q]
            self.q += 1
            return v

    class Margins(PyNUTS):
        best = (np.inf, "", 0)
        def note(self, margin, kind):
            if margin < self.best[0]:
                self.best = (float(margin), kind, self.nleap)
        def build_tree(self, x, r, grad, logu, direction, depth, phi):
            out = super().build_tree(x, r, grad, logu, direction, depth, phi)
            if depth == 0:
                xq, rq = out[6], out[7]
                joint = self.target.logpdf(xq, phi) - 0.5 * np.dot(rq, rq)
                scale = max(1.0, abs(joint), abs(logu))
                self.note(abs(logu - joint) / scale, "slice test")
                self.note(abs(logu - 100.0 - joint) / scale, "divergence test")
            return out
        def stop_criterion(self, xm, xp, rm, rp):
            dx = xp - xm
            for rr in (rm, rp):
                d = float(np.dot(dx, rr.T))
                self.note(abs(d) / max(np.sum(np.abs(dx * rr)), 1e-300), "U-turn product")
            return PyNUTS.stop_criterion(xm, xp, rm, rp)

    nuts = Margins(ot, eps, PhiloxRNG())
    nuts.generate_nuts_samples(np.asarray(x0, dtype=np.float64), np.asarray(r0, dtype=np.float64), 1.0)
    return nuts.best


@pytest.mark.parametrize("T,eps", [(137, 0.01), (1, 0.05), (200, 0.002), (437, 0.01), (9, 0.02)])
def test_arma_other_series_lengths_and_deep_trees(tmp_path, T, eps):
    """arma with other series lengths than the shipped 200 (the lane kernel takes any T: shorter, longer --
    StanModel("arma", data_path=...) never leaves the device path -- and not a multiple of its 8- and 16-step
    chunks) and, at the full length with a small step, trees of depth 5-8, which reach the tree-stack levels
    kept in LDS and in the global overflow area."""
    import json
    from smcnuts_amd import ArmaModel, _capi
    src = json.load(open(os.path.join(DATA, "arma.json")))
    path = str(tmp_path / "arma_T.json")
    yy = (src["y"] * 3)[:T] if T > 200 else src["y"][:T]
    json.dump({"T": T, "y": yy}, open(path, "w"))
    t, ot = ArmaModel(path), orc.OracleTarget(orc.MODEL_ARMA, orc.arma_data(path), 4)
    N, seed = 4096, 99
    x = np.random.default_rng(T).normal(size=(N, 4)) * np.array([0.05, 0.05, 0.1, 0.1]) + np.array([0, 0.9, 0, -1.8])
    close(t.logpdf(x, 0.7), ot.logpdf(x, 0.7), rtol=2e-13, atol=2e-12)
    close(t.logpdfgrad(x, 0.7), ot.logpdfgrad(x, 0.7), rtol=1e-12, atol=1e-11)
    ctx = _capi.Context(N, t.model_id, t.model_data)
    ctx.set_seed(seed)
    ctx.set_state(x=x, logw=np.zeros(N))
    ctx.propose_nuts(eps, 1.0, 3)
    r, xn, rn, _ = ctx.get_proposal()
    st = ctx.tree_stats()
    ref = orc.nuts_rvs(ot, x, r, 1.0, eps, seed=seed, iteration=3)
    mism = np.flatnonzero(st["ndraws"] != ref["ndraws"])
    # A particle may take another tree than the oracle's ONLY through a comparison that is a tie to rounding (device
    # exp / log1p against libm, fused multiply-adds): T = 1 is the prior alone, whose trees run for hundreds of leapfrogs
    # along near-periodic orbits and graze their U-turn criterion.  The allowance is per case; every mismatching particle
    # must SHOW its tie: the smallest relative margin of any slice / divergence / U-turn comparison along the oracle's
    # own tree (re-built by the instrumented Python restatement below) is at rounding level, and the message names it.
    allowed = {1: 3}.get(T, 0)
    ties = [_closest_comparison(ot, x[i], r[i], eps, seed, 3, int(i)) for i in mism]
    report = "; ".join(f"particle {int(i)}: {kind} decided by {m:.1e} (relative) at leapfrog {at}" for i, (m, kind, at) in zip(mism, ties))
    assert mism.size <= allowed, f"T={T}: particles {mism.tolist()} took a different tree (allowed {allowed}): {report}"
    for i, (m, kind, at) in zip(mism, ties):
        assert m < 1e-9, f"T={T}: particle {int(i)} differs from the oracle without a tie in the oracle's tree ({report})"
    if mism.size:
        print(f"T={T}: {report}")
    ok = np.setdiff1d(np.arange(N), mism)
    np.testing.assert_array_equal(st["depth"][ok], ref["depth"][ok])
    close(xn[ok], ref["x_new"][ok], rtol=1e-11, atol=1e-12)
    if T == 200:
        assert (st["depth"] >= 6).sum() > 100      # the overflow levels were exercised


def test_full_size_properties_arma_65536():
    """BASELINE config 2 size (N = 65 536): size-independent properties.
    Determinism (same seed twice => bit-identical), weights normalised,
    resampling indices sorted-CDF consistent, posterior means near
    stan_models/arma/arma.params."""
    from smcnuts_amd import ArmaModel, SMCSampler
    truth = np.array([0.00678443422162953, 0.9570083053800078, -0.03407898212798232, 0.1666098193000008])
    outs = []
    for _ in range(2):
        smc = SMCSampler(K=12, N=65536, target=ArmaModel(), step_size=0.01, seed=99, save_history=False)
        smc.sample(show_progress=False)
        outs.append((smc.mean_estimate.copy(), smc.ess.copy(), smc.samples.ctx.get_state()[0], smc.leapfrogs.copy()))
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)
    wn = smc.samples.wn
    close(wn.sum(), 1.0, rtol=1e-14)
    assert np.all(np.abs(outs[0][0][-1] - truth) < np.array([0.002, 0.004, 0.01, 0.002]))
    assert smc.leapfrogs.sum() > 65536 * 12 * 3


@pytest.mark.parametrize("name", ["gauss4_fwd", "gauss32_fwd", "arma_fwd", "gauss4_deep", "gauss256_fwd"])
def test_device_resident_loop_on_reference_draws(golden_dir, name):
    """The device-resident loop (no host round trip per iteration; resample
    decision, estimates and counters on the device; shifted one-pass variance)
    replaying the reference's draws."""
    from smcnuts_amd import SMCSampler
    g = load(golden_dir, name)
    t, _ = targets(name)
    K, N = int(g["K"]), int(g["N"])
    smc = SMCSampler(K=K, N=N, target=t, step_size=float(g["eps"]), lkernel="forwardsLKernel",
                     x0=g["x0"], logq0=g["logq0"], seed=1)
    assert smc.device_resident
    for k in range(K):
        u = g[f"u_resample_{k}"]
        smc.step_async(tape=g[f"tape_{k}"], tape_off=g[f"tape_off_{k}"], r=g[f"r_{k}"],
                       u_resample=u if u.size else None)
    smc.finalise_async()
    for k in range(K):
        assert bool(smc.resampled[k]) == bool(g[f"resampled_{k}"])
    close(smc.x_saved, g["x_saved"], rtol=1e-12, atol=1e-13)
    close(smc.logw_saved, g["logw_saved"], rtol=1e-12, atol=1e-11)
    close(smc.ess, g["ess"], rtol=1e-11)
    close(smc.log_likelihood, g["log_likelihood"], rtol=1e-12, atol=1e-12)
    close(smc.mean_estimate, g["mean_estimate"], rtol=5e-11, atol=5e-13)
    close(smc.variance_estimate, g["variance_estimate"], rtol=1e-10, atol=1e-13)
    close(smc.acceptance_rate, g["acceptance_rate"], atol=1e-12)
    close(smc.phi, g["phi"])


@pytest.mark.parametrize("wide", [False, True])
def test_device_resident_equals_stepwise_philox(wide):
    """Same seed: sample() (device-resident) and the step-by-step loop agree;
    states bit for bit, scalars to reduction round-off.  With the wide evaluation on (smcn_set_wide_eval: lane groups
    take over a wavefront's last stragglers, WHICH evaluations depends on the launch's schedule) the two drivers run the
    same trees and agree to rounding instead."""
    from smcnuts_amd import ArmaModel, SMCSampler
    a = SMCSampler(K=8, N=4096, target=ArmaModel(), step_size=0.01, seed=3, wide_eval=wide)
    a.sample(show_progress=False)
    b = SMCSampler(K=8, N=4096, target=ArmaModel(), step_size=0.01, seed=3, wide_eval=wide)
    for _ in range(8):
        b.step()
    b.finalise()
    assert a.resampled == b.resampled and any(a.resampled)
    if wide:
        same_run_to_rounding(a, b, "sample() against the step-by-step loop")
        return
    np.testing.assert_array_equal(a.x_saved, b.x_saved)
    np.testing.assert_array_equal(a.leapfrogs, b.leapfrogs)
    close(a.logw_saved, b.logw_saved, rtol=1e-14, atol=1e-15)
    close(a.ess, b.ess, rtol=1e-13)
    close(a.mean_estimate, b.mean_estimate, rtol=1e-13, atol=1e-15)
    close(a.variance_estimate, b.variance_estimate, rtol=5e-9, atol=5e-15)
    close(a.acceptance_rate, b.acceptance_rate)


def test_samplers_built_on_reused_streams_and_buffers_run_the_same():
    """The library pools its streams and device buffers per device (smcn_api.hip: pool_take / cached_malloc -- a context's
    buffers go back to a cache when it is closed and are handed out again, zeroed, to a request of the same size).  A
    sampler built on another one's buffers must be the sampler built on fresh memory: the same seed gives the same run
    bit for bit, again and again, with differently sized samplers (whose buffers cannot be reused) and with a second
    live sampler in between."""
    from smcnuts_amd import ArmaModel, IsoGaussian, SMCSampler

    def arma(seed, n=4096, k=8):
        s = SMCSampler(K=k, N=n, target=ArmaModel(), step_size=0.01, seed=seed, wide_eval=False)
        s.sample(show_progress=False)
        out = (s.x_saved.copy(), s.logw_saved.copy(), s.ess.copy(), s.mean_estimate.copy(), s.leapfrogs.copy())
        s.samples.ctx.close()
        return out

    first = arma(5)
    other = SMCSampler(K=4, N=2048, target=IsoGaussian(32), step_size=0.2, seed=1)      # stays alive across the repeats
    other.sample(show_progress=False)
    for rep in range(4):
        if rep % 2:
            arma(6, n=4160, k=5)                                  # other sizes in between: nothing of it fits the cache's slots
        again = arma(5)
        for a, b in zip(first, again):
            np.testing.assert_array_equal(a, b)
    g1 = other.mean_estimate.copy()
    other2 = SMCSampler(K=4, N=2048, target=IsoGaussian(32), step_size=0.2, seed=1)
    other2.sample(show_progress=False)
    np.testing.assert_array_equal(other2.mean_estimate, g1)
    # the idle buffers go back to the driver on request (smcn_device_cache_trim); a sampler built after that is the same
    from smcnuts_amd import _capi
    released, idle = _capi.trim_device_cache()
    assert idle > 0 and released == idle
    assert _capi.trim_device_cache() == (0, 0)
    for a, b in zip(first, arma(5)):
        np.testing.assert_array_equal(a, b)


def test_preallocate_false_runs_the_same():
    """SMCSampler(preallocate=False) leaves the device-resident loop's buffers (history, block partials, transition records) to
    the first sample(): the same run, bit for bit, and a host-loop-only caller never allocates them."""
    from smcnuts_amd import ArmaModel, SMCSampler

    def run(pre):
        smc = SMCSampler(K=6, N=4096, target=ArmaModel(), step_size=0.01, seed=31, preallocate=pre)
        started = smc._fast_started
        smc.sample(show_progress=False)
        return smc, started

    a, sa = run(True)
    b, sb = run(False)
    assert sa and not sb
    for name in ("x_saved", "logw_saved", "ess", "mean_estimate", "variance_estimate", "leapfrogs", "log_likelihood"):
        np.testing.assert_array_equal(getattr(a, name), getattr(b, name), err_msg=name)
    c = SMCSampler(K=2, N=1024, target=ArmaModel(), step_size=0.01, seed=31, preallocate=False)
    c.step(); c.step(); c.finalise()
    assert not c._fast_started and np.all(np.isfinite(c.mean_estimate))


def test_device_side_bisection_equals_the_host_driven_one():
    """ESSTempering.calculate_phi (adaptive_tempering.py:18-63): the bisection that runs on the device (four steps of
    scipy's bisect.c per pass, one host wait per SMC iteration) returns the temperatures of the host-driven loop (one
    reduction and one wait per trial point) -- both restate bisect.c, so they walk the same midpoints unless a trial
    point's ESS - N/2 is zero to rounding -- for arma (tempering + forward L-kernel) and PRMwCD (Gaussian L-kernel)."""
    from smcnuts_amd import ArmaModel, PRMwCDModel, SMCSampler
    from smcnuts_amd.tempering.adaptive_tempering import ESSTempering
    for mk, kw in ((ArmaModel, dict(K=8, N=4096, step_size=0.01, seed=5, lkernel="forwardsLKernel", tempering=True)),
                   (PRMwCDModel, dict(K=6, N=2048, step_size=0.01, seed=9, lkernel="GaussianApproxLKernel", tempering=True))):
        runs = []
        for dev in (True, False):
            ESSTempering.device_bisection = dev
            try:
                s = SMCSampler(target=mk(), wide_eval=False, **kw)
                s.sample(show_progress=False)
            finally:
                ESSTempering.device_bisection = True
            runs.append(s)
        a, b = runs
        assert 0.0 < a.phi[0] < 1.0 and np.all(np.diff(a.phi) >= 0)
        close(a.phi, b.phi, rtol=0, atol=1e-14)
        close(a.ess, b.ess, rtol=1e-10)
    # the same question asked directly on resident density parts (particles from N(0, I): the heaviest one sits anywhere
    # in the population), for several population sizes and brackets
    import ctypes as C
    from smcnuts_amd import _capi
    from smcnuts_amd.parallel import combine_lse_partials
    from smcnuts_amd.tempering.adaptive_tempering import bisect
    for N in (64, 1000, 1024, 20000):
        t = ArmaModel()
        ctx = _capi.Context(N, t.model_id, t.model_data)
        ctx.set_seed(N)
        x = np.random.default_rng(N).normal(size=(N, 4)) * np.array([0.3, 0.3, 0.3, 0.5]) + np.array([0, 0.9, 0, -1.8])
        ctx.set_state(x=x, logw=np.zeros(N))
        ctx.call("smcn_eval_proposed_parts", 0)
        for po in (0.0, 1e-4, 0.3):
            def f(phi):
                _, sw = combine_lse_partials(ctx.temper_partials(po, phi)[None, :])
                return 1.0 / sw - 0.5 * N
            want = 1.0 if f(1.0) >= 0 else bisect(f, po, 1.0)
            phi, st = C.c_double(0.0), C.c_int(9)
            ctx.call("smcn_temper_bisect", po, 0.5 * N, C.byref(phi), C.byref(st))
            assert st.value == 0
            close(phi.value, want, rtol=0, atol=1e-14, err_msg=f"N={N} phi_old={po}")


@pytest.mark.parametrize("D", [4, 13, 32])
def test_gaussian_lkernel_algebra_on_the_device(D):
    """smcn_gauss_lkernel_device: the D x D algebra of the Gaussian L-kernel (gaussian_lkernel.py:45-82: pinv of c_xx, the
    conditional covariance with its ridge, the normal log-density) by two Cholesky factorisations of one wavefront gives the
    L values of the NumPy path (the reference's own pinv / eigh calls) and of the oracle's literal restatement -- and
    refuses, handing the decision back to those calls, when a covariance is singular."""
    from smcnuts_amd import GaussianTarget, SMCSampler, _capi
    from smcnuts_amd.lkernel.gaussian_lkernel import GaussianApproxLKernel
    N = 4096
    smc = SMCSampler(K=2, N=N, target=GaussianTarget(D), step_size=0.05, lkernel="GaussianApproxLKernel", seed=5)
    s = smc.samples
    assert isinstance(s.lkernel, GaussianApproxLKernel)
    rng = np.random.default_rng(D)
    A = rng.standard_normal((D, D)) / np.sqrt(D) + np.eye(D)
    x_new = rng.standard_normal((N, D)) @ A + rng.standard_normal(D)
    r_new = 0.4 * x_new @ rng.standard_normal((D, D)) / np.sqrt(D) + rng.standard_normal((N, D)) * rng.uniform(0.5, 2.0, D)
    r = rng.standard_normal((N, D))

    def logw_new(device):
        s.ctx.call("smcn_set_proposal", *(_capi.dptr(np.ascontiguousarray(a)) for a in (r, x_new, r_new)))
        s.lkernel.device_algebra = device
        try:
            s.reweight()
        finally:
            s.lkernel.device_algebra = True
        return s.ctx.get_proposal(r=False, x_new=False, r_new=False, logw_new=True)[3], s.lkernel.last_path

    dev, path_d = logw_new(True)
    host, path_h = logw_new(False)
    assert (path_d, path_h) == ("device", "host")
    assert np.all(np.isfinite(dev))
    close(dev, host, rtol=0, atol=1e-11 * max(1.0, np.abs(host).max()))
    # (the host path against the reference's literal formulation and the oracle: tests/test_host_logic.py)
    # a degenerate population (every x' the same point in one coordinate): c_xx is singular, pinv's cut-off decides -- on the host
    x_deg = x_new.copy()
    x_deg[:, 0] = 1.25
    x_new_keep, x_new = x_new, x_deg
    deg, path = logw_new(True)
    x_new = x_new_keep
    assert path == "host" and np.all(np.isfinite(deg))


@pytest.mark.parametrize("lanes", [64, 32, 16, 8, 4])
@pytest.mark.parametrize("T", [200, 64, 137, 383])
def test_wide_evaluation_equals_one_lane(tmp_path, lanes, T):
    """smcn_set_wide_eval: the arma recurrence cut into 64 / 32 / 16 / 4 segments over a lane group (state pass, scan of
    the affine segment maps -- within DPP rows, then across them --, full pass, butterfly) gives the four sums of the
    one-lane loop to rounding, for series lengths that do and do not divide into the segments, owners scattered over the
    wavefront, and theta near +-1 (slowly decaying segment coupling) as well as near 0."""
    if lanes >= 32 and T < 130:
        pytest.skip("32 / 64 lanes per series take at least 130 observations (every segment two steps)")
    import json
    from smcnuts_amd import ArmaModel, _capi
    src = json.load(open(os.path.join(DATA, "arma.json")))
    path = str(tmp_path / "arma_T.json")
    json.dump({"T": T, "y": (src["y"] * 2)[:T]}, open(path, "w"))
    t = ArmaModel(path)
    rng = np.random.default_rng(T + lanes)
    M = 1003
    x = rng.normal(size=(M, 4)) * np.array([0.05, 0.05, 0.1, 0.1]) + np.array([0, 0.9, 0, -1.8])
    x[:200, 2] = rng.uniform(-0.999, 0.999, size=200)          # theta over the whole stationary range
    x[200:220, 2] = rng.uniform(1.0, 1.05, size=20) * rng.choice([-1, 1], size=20)   # and slightly beyond (growing modes)
    ctx = _capi.Context(64, t.model_id, t.model_data)
    out = np.zeros((M, 8))
    ctx.call("smcn_selftest_wide", lanes, _capi.dptr(np.ascontiguousarray(x)), M, _capi.dptr(out))
    one, wide = out[:, :4], out[:, 4:]
    assert np.all(np.isfinite(one)) and np.all(np.isfinite(wide))
    scale = np.abs(one).max(axis=1, keepdims=True) + 1.0        # the sensitivity sums cancel against each other
    assert np.max(np.abs(one - wide) / scale) < 1e-12, np.max(np.abs(one - wide) / scale)


def test_device_math():
    """exp_fast / log1p_pos / rcp_nr (smcn_device.hpp) against libm: <= 4 ulp."""
    from smcnuts_amd import GaussianTarget, _capi
    ctx = _capi.Context(256, 0, GaussianTarget(2).model_data)
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-700, 700, 20000), rng.uniform(-8, 2, 20000), rng.normal(size=20000) * 1e-3,
                        10.0 ** rng.uniform(-300, 300, 20000), [0.0, 1.0, -1.0, 0.5 * np.log(2), 1e-320]])
    out = np.empty(11 * x.size)
    ctx.call("smcn_selftest_math", _capi.dptr(np.ascontiguousarray(x)), x.size, _capi.dptr(out))
    e, l, r = out[:x.size], out[x.size:2 * x.size], out[2 * x.size:3 * x.size]
    # the wavefront butterfly: v_permlane16/32_swap stages == ds_bpermute stages, bit for bit, and == the same
    # pairing order summed on the host
    s_swap, s_perm = out[3 * x.size:4 * x.size], out[4 * x.size:5 * x.size]
    np.testing.assert_array_equal(s_swap, s_perm)
    xp = np.concatenate([x, np.zeros(-x.size % 64)]).reshape(-1, 64)
    with np.errstate(all="ignore"):
        v = xp.copy()
        for mask in (1, 2):                       # xor 1, xor 2
            v = v + v[:, np.arange(64) ^ mask]
        v = v + v[:, (np.arange(64) & ~7) + 7 - (np.arange(64) & 7)]        # row_half_mirror
        v = v + v[:, (np.arange(64) & ~15) + 15 - (np.arange(64) & 15)]     # row_mirror
        for mask in (16, 32):
            v = v + v[:, np.arange(64) ^ mask]
    np.testing.assert_array_equal(s_swap, v.reshape(-1)[:x.size])
    # the fused butterflies (two / four sums in one: v_permlane32_swap, v_permlane16_swap with DIFFERENT operands, then the
    # row stages): halves first, then rows, then xor 1, xor 2, half mirror, mirror -- the same bits as that order on the host
    def fold(w):
        with np.errstate(all="ignore"):
            w = w + w[:, np.arange(64) ^ 32]
            w = w + w[:, np.arange(64) ^ 16]
            for mask in (1, 2):
                w = w + w[:, np.arange(64) ^ mask]
            w = w + w[:, (np.arange(64) & ~7) + 7 - (np.arange(64) & 7)]
            w = w + w[:, (np.arange(64) & ~15) + 15 - (np.arange(64) & 15)]
        return w[:, :1].repeat(64, axis=1).reshape(-1)[:x.size]
    n = x.size
    with np.errstate(all="ignore"):
        cols = [xp, xp * xp, np.abs(xp), 1.0 - xp]
    for k, col in enumerate(cols[:2]):
        np.testing.assert_array_equal(out[(5 + k) * n:(6 + k) * n], fold(col))
    for k, col in enumerate(cols):
        np.testing.assert_array_equal(out[(7 + k) * n:(8 + k) * n], fold(col))
    with np.errstate(all="ignore"):
        ok = np.abs(x) < 700
        close(e[ok], np.exp(x[ok]), rtol=9e-16)
        close(l, np.log1p(np.abs(x)), rtol=9e-16, atol=1e-320)
        nz = (np.abs(x) > 1e-300) & (np.abs(x) < 1e300)
        close(r[nz], 1.0 / x[nz], rtol=5e-16)


def test_config5_shape_d256_philox_vs_oracle():
    """BASELINE config 5 target (iso-Gaussian D = 256, HBM tree stack) at a size
    that queues several particles per resident wavefront; Philox on both sides."""
    from smcnuts_amd import IsoGaussian, _capi
    N, D, seed = 12000, 256, 17
    t = IsoGaussian(D)
    ot = orc.OracleTarget(orc.MODEL_GAUSS, orc.gauss_data(D), D)
    x = np.random.default_rng(2).normal(size=(N, D))
    ctx = _capi.Context(N, t.model_id, t.model_data)
    ctx.set_seed(seed)
    ctx.set_state(x=x, logw=np.zeros(N))
    ctx.propose_nuts(0.25, 1.0, 4)
    r, xn, rn, _ = ctx.get_proposal()
    st = ctx.tree_stats()
    ref = orc.nuts_rvs(ot, x, r, 1.0, 0.25, seed=seed, iteration=4)
    np.testing.assert_array_equal(st["ndraws"], ref["ndraws"])
    np.testing.assert_array_equal(st["nleap"], ref["nleap"])
    close(xn, ref["x_new"], rtol=1e-13, atol=1e-14)
    close(rn, ref["r_new"], rtol=1e-13, atol=1e-14)


@pytest.mark.parametrize("D", [100, 256])
def test_wave_kernel_momentum_layouts_convert_in_place(D):
    """Targets whose particle fills a wavefront keep the momentum PARTICLE-MAJOR (smcn_ctx::r_pm: the NUTS kernel reads and
    writes a particle's row as coalesced pieces).  smcn_get_proposal hands r and r' out from that layout; a kernel that wants
    [D][N] -- here the re-weighting by passes over r, r' (smcn_reweight) -- first converts both in place, after which the
    same call hands out the same arrays from the other layout; and the re-weighting's log q - log L (samples.py:193-194)
    from the converted arrays equals the one from the downloaded ones."""
    from smcnuts_amd import GaussianTarget, _capi
    N, seed = 3000, 23
    t = GaussianTarget(D)
    x = np.random.default_rng(4).normal(size=(N, D))
    ctx = _capi.Context(N, t.model_id, t.model_data)
    ctx.set_seed(seed)
    ctx.set_state(x=x, logw=np.zeros(N))
    ctx.propose_nuts(0.2, 1.0, 2)
    r1, xn1, rn1, _ = ctx.get_proposal()                      # particle-major on the device: plain copies
    ctx.call("smcn_reweight", _capi.LKERNEL_FORWARD)          # reads r, r' as [D][N]: converts them in place first
    r2, xn2, rn2, lw = ctx.get_proposal(logw_new=True)        # [D][N] on the device: transposed downloads
    np.testing.assert_array_equal(r1, r2)
    np.testing.assert_array_equal(rn1, rn2)
    np.testing.assert_array_equal(xn1, xn2)
    lp0, ll0, lp1, ll1 = ctx.density_parts()
    want = (lp1 + ll1) - (lp0 + ll0) + (-0.5 * (rn1 ** 2).sum(axis=1)) - (-0.5 * (r1 ** 2).sum(axis=1))
    close(lw, want, rtol=1e-12, atol=1e-10)
    ctx.propose_nuts(0.2, 1.0, 3)                             # and the next launch draws particle-major momenta again
    r3, _, rn3, _ = ctx.get_proposal()
    assert np.abs(r3 - r1).max() > 0.5 and np.all(np.isfinite(rn3))
    close(r3.std(), 1.0, rtol=2e-2)


@pytest.mark.parametrize("D,phi", [(100, 0.4), (256, 0.4), (300, 1.0), (512, 0.7)])
def test_wave_kernel_with_a_likelihood_factor_vs_oracle(D, phi):
    """nuts_wave_kernel's other instantiations: a target WITH a likelihood factor (prior N(0, 3^2 I) x N(x | 0.3, 0.8^2 I), at a
    temperature phi) and dimensions that do / do not fill the lanes' slots (100: masked, 4 per lane; 256: full; 300: masked,
    8 per lane; 512: full, 8 per lane); Philox on both sides -- draws consumed, leapfrog counts and depths exact, states and
    density parts to rounding."""
    from smcnuts_amd import GaussianTarget, _capi
    N, seed = 2500, 29
    t = GaussianTarget(D, prior_sd=3.0, lik_mean=0.3, lik_sd=0.8)
    ot = orc.OracleTarget(orc.MODEL_GAUSS, orc.gauss_data(D, prior_sd=3.0, lik_mean=0.3, lik_sd=0.8), D)
    x = 0.3 + 0.9 * np.random.default_rng(6).normal(size=(N, D))
    ctx = _capi.Context(N, t.model_id, t.model_data)
    ctx.set_seed(seed)
    ctx.set_state(x=x, logw=np.zeros(N))
    ctx.propose_nuts(0.05, phi, 5)
    r, xn, rn, _ = ctx.get_proposal()
    st = ctx.tree_stats()
    ref = orc.nuts_rvs(ot, x, r, phi, 0.05, seed=seed, iteration=5)
    np.testing.assert_array_equal(st["ndraws"], ref["ndraws"])
    np.testing.assert_array_equal(st["nleap"], ref["nleap"])
    np.testing.assert_array_equal(st["depth"], ref["depth"])
    assert st["nleap"].max() >= 15
    close(xn, ref["x_new"], rtol=1e-12, atol=1e-13)
    close(rn, ref["r_new"], rtol=1e-12, atol=1e-13)
    lp0, ll0, lp1, ll1 = ctx.density_parts()
    close(lp1, ref["lpri1"], rtol=1e-12, atol=1e-11)
    close(ll1, ref["llik1"], rtol=1e-12, atol=1e-11)
    close(lp0, ref["lpri0"], rtol=1e-12, atol=1e-11)


def test_wave_kernel_on_particles_without_a_finite_density():
    """The target adapter's failure convention (bridgestan.py:45-49, 79-80: a non-finite density is -inf with a gradient of
    -inf) inside nuts_wave_kernel: a particle that STARTS at a NaN / overflowing position, and one whose first leapfrog
    leaves the representable range, take the oracle's trees (draws, leapfrogs, depth) and stay where the oracle leaves them."""
    from smcnuts_amd import IsoGaussian, _capi
    N, D, seed = 512, 256, 41
    t = IsoGaussian(D)
    ot = orc.OracleTarget(orc.MODEL_GAUSS, orc.gauss_data(D), D)
    x = np.random.default_rng(8).normal(size=(N, D))
    x[3, 7] = np.nan
    x[11, 0] = 1e200                      # x^2 overflows: log pi = -inf at the start
    x[19, 5] = 1e154                      # finite density at the start, gradient 1e154: the first leaf overflows
    ctx = _capi.Context(N, t.model_id, t.model_data)
    ctx.set_seed(seed)
    ctx.set_state(x=x, logw=np.zeros(N))
    ctx.propose_nuts(0.25, 1.0, 1)
    r, xn, rn, _ = ctx.get_proposal()
    st = ctx.tree_stats()
    ref = orc.nuts_rvs(ot, x, r, 1.0, 0.25, seed=seed, iteration=1)
    np.testing.assert_array_equal(st["ndraws"], ref["ndraws"])
    np.testing.assert_array_equal(st["nleap"], ref["nleap"])
    np.testing.assert_array_equal(st["depth"], ref["depth"])
    for p in (3, 11, 19):
        np.testing.assert_array_equal(xn[p], ref["x_new"][p])       # (NaN == NaN here)
    ok = np.setdiff1d(np.arange(N), [3, 11, 19])
    close(xn[ok], ref["x_new"][ok], rtol=1e-13, atol=1e-14)


def test_gaussian_beyond_256_dimensions_vs_oracle():
    """D = 300 (8 coordinates per lane, tree stack in HBM): Philox on both sides, decisions exact."""
    from smcnuts_amd import IsoGaussian, _capi
    N, D, seed = 3000, 300, 23
    t = IsoGaussian(D)
    ot = orc.OracleTarget(orc.MODEL_GAUSS, orc.gauss_data(D), D)
    x = np.random.default_rng(3).normal(size=(N, D))
    ctx = _capi.Context(N, t.model_id, t.model_data)
    ctx.set_seed(seed)
    ctx.set_state(x=x, logw=np.zeros(N))
    ctx.propose_nuts(0.2, 1.0, 2)
    r, xn, rn, _ = ctx.get_proposal()
    st = ctx.tree_stats()
    ref = orc.nuts_rvs(ot, x, r, 1.0, 0.2, seed=seed, iteration=2)
    np.testing.assert_array_equal(st["ndraws"], ref["ndraws"])
    np.testing.assert_array_equal(st["nleap"], ref["nleap"])
    close(xn, ref["x_new"], rtol=2e-13, atol=2e-14)
    close(t.logpdf(x[:50]), ot.logpdf(x[:50]), rtol=1e-14)


@pytest.mark.parametrize("N", [1, 3, 9, 17, 33, 1000])
def test_wide_particles_ragged_and_tiny_populations(N):
    """The wave-per-particle kernel deals cache lines of 8 particles to per-XCD queues (as many queues as the grid has
    blocks, 8 at most): populations of fewer particles than queues, ragged last lines and partial blocks all finish and
    take the oracle's trees."""
    from smcnuts_amd import IsoGaussian, _capi
    D, seed = 200, 5
    t = IsoGaussian(D)
    ot = orc.OracleTarget(orc.MODEL_GAUSS, orc.gauss_data(D), D)
    x = np.random.default_rng(N).normal(size=(N, D))
    ctx = _capi.Context(N, t.model_id, t.model_data)
    ctx.set_seed(seed)
    ctx.set_state(x=x, logw=np.zeros(N))
    ctx.propose_nuts(0.15, 1.0, 1)
    r, xn, rn, _ = ctx.get_proposal()
    st = ctx.tree_stats()
    ref = orc.nuts_rvs(ot, x, r, 1.0, 0.15, seed=seed, iteration=1)
    np.testing.assert_array_equal(st["ndraws"], ref["ndraws"])
    np.testing.assert_array_equal(st["nleap"], ref["nleap"])
    assert st["nleap"].min() > 0
    close(xn, ref["x_new"], rtol=2e-13, atol=2e-14)
    close(rn, ref["r_new"], rtol=2e-13, atol=2e-14)


def test_config5_per_gpu_size_properties():
    """BASELINE config 5 at its per-GPU size (iso-Gaussian D = 256, N = 131 072, tree stack in HBM):
    run-to-run determinism, normalised weights, and the N(0, I) moments of the target (every coordinate's
    weighted mean within 5 standard errors of 0, variance near 1)."""
    from smcnuts_amd import IsoGaussian, SMCSampler
    N, D, K = 131072, 256, 3
    outs = []
    for _ in range(2):
        smc = SMCSampler(K=K, N=N, target=IsoGaussian(D), step_size=0.25, seed=77, save_history=False)
        smc.sample(show_progress=False)
        outs.append((smc.mean_estimate.copy(), smc.variance_estimate.copy(), smc.ess.copy(), smc.leapfrogs.copy()))
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)
    mean, var, ess, leaps = outs[0]
    assert np.all(np.isfinite(mean)) and np.all(ess > 0.9 * N)           # x0 ~ N(0, I) = the target: weights stay flat
    assert np.all(np.abs(mean[-1]) < 5.0 / np.sqrt(N)) and np.all(np.abs(var[-1] - 1.0) < 0.03)
    assert leaps.min() > 5 * N and abs(smc.samples.wn.sum() - 1.0) < 1e-12


@pytest.mark.parametrize("fuse_max,wide", [(1, False), (3, False), (8, False), (64, False), (8, True), (64, True)])
def test_fused_transitions_equal_one_launch_per_iteration(fuse_max, wide):
    """Several SMC iterations per NUTS launch (speculating "no resampling", rolled back
    when a generation has to resample) reproduce the one-launch-per-iteration loop bit
    for bit: the chain below resamples at iterations 0, 1 and later again.  (wide: the lane-group evaluation of
    stragglers re-associates the likelihood sums depending on the schedule -- same trees, states to rounding.)"""
    from smcnuts_amd import ArmaModel, SMCSampler
    K, N = 30, 256          # few particles: the ESS crosses N/2 several times along the chain
    for seed in range(3, 12):
        a = SMCSampler(K=K, N=N, target=ArmaModel(), step_size=0.01, seed=seed, wide_eval=wide)
        for _ in range(K):
            a.step_async()
        a.finalise_async()
        if sum(a.resampled[2:]) >= 2:      # resampling in the middle of would-be fused blocks
            break
    else:
        pytest.fail("no seed gave a chain that resamples mid-way")
    b = SMCSampler(K=K, N=N, target=ArmaModel(), step_size=0.01, seed=seed, wide_eval=wide)
    b.run_fused(fuse_max=fuse_max)
    b.finalise_async()
    assert a.resampled == b.resampled
    if wide:
        same_run_to_rounding(a, b, f"fused blocks of up to {fuse_max} against one launch per iteration")
        return
    np.testing.assert_array_equal(a.leapfrogs, b.leapfrogs)
    np.testing.assert_array_equal(a.acceptance_rate, b.acceptance_rate)
    np.testing.assert_array_equal(a.x_saved, b.x_saved)
    np.testing.assert_array_equal(a.logw_saved, b.logw_saved)
    close(a.ess, b.ess, rtol=1e-14)
    close(a.log_likelihood, b.log_likelihood, rtol=1e-14)
    close(a.mean_estimate, b.mean_estimate, rtol=1e-14, atol=1e-16)
    close(a.variance_estimate, b.variance_estimate, rtol=1e-11, atol=1e-16)


def test_block_size_follows_the_ess_trend():
    """The pipelined driver sizes its speculation from the decay of the ESS: a slowly decaying
    population (arma at N = 8192, after the initial resamplings) is advanced in few, long blocks
    and the results still equal the one-launch-per-iteration chain."""
    from smcnuts_amd import ArmaModel, SMCSampler
    K, N, seed = 60, 8192, 10
    a = SMCSampler(K=K, N=N, target=ArmaModel(), step_size=0.01, seed=seed, save_history=False, wide_eval=False)
    for _ in range(K):
        a.step_async()
    a.finalise_async()
    b = SMCSampler(K=K, N=N, target=ArmaModel(), step_size=0.01, seed=seed, save_history=False, wide_eval=False)
    b.samples.ctx.timers(reset=True)
    b.run_fused(fuse_max=64)
    launches = b.samples.ctx.timers()[1]
    b.finalise_async()
    assert a.resampled == b.resampled
    np.testing.assert_array_equal(a.leapfrogs, b.leapfrogs)
    close(a.ess, b.ess, rtol=1e-14)
    close(a.mean_estimate, b.mean_estimate, rtol=1e-14, atol=1e-16)
    last = max(i for i, r in enumerate(a.resampled) if r)
    assert launches <= last + 1 + 8, (launches, last)    # a handful of launches for the K - last clean iterations


def test_fused_without_history_and_late_resampling():
    """save_history=False uses the generation ring; a Gaussian-target chain with a
    wide prior resamples late (after several clean fused blocks)."""
    from smcnuts_amd import GaussianTarget, ArmaModel, SMCSampler
    for tgt, eps in ((ArmaModel(), 0.01),):
        a = SMCSampler(K=20, N=8192, target=tgt, step_size=eps, seed=11, save_history=False, wide_eval=False)
        for _ in range(20):
            a.step_async()
        a.finalise_async()
        b = SMCSampler(K=20, N=8192, target=tgt, step_size=eps, seed=11, save_history=False, wide_eval=False)
        b.run_fused(fuse_max=8)
        b.finalise_async()
        assert a.resampled == b.resampled
        np.testing.assert_array_equal(a.samples.x, b.samples.x)
        np.testing.assert_array_equal(a.leapfrogs, b.leapfrogs)
        close(a.ess, b.ess, rtol=1e-14)
        close(a.mean_estimate, b.mean_estimate, rtol=1e-14, atol=1e-16)


@pytest.mark.parametrize("N", [1, 7, 100, 1025, 3001])
def test_ragged_particle_counts(N):
    """Shard sizes that are no multiple of the group, wavefront, block or scan-tile
    size: NUTS + weight path against the oracle (Philox)."""
    from smcnuts_amd import ArmaModel, _capi
    t = ArmaModel()
    ot = orc.OracleTarget(orc.MODEL_ARMA, t.model_data, 4)
    rng = np.random.default_rng(N)
    x = rng.normal(size=(N, 4)) * np.array([0.05, 0.05, 0.1, 0.1]) + np.array([0, 0.9, 0, -1.8])
    logw = rng.normal(size=N) * 2
    ctx = _capi.Context(N, t.model_id, t.model_data)
    ctx.set_seed(9)
    ctx.set_state(x=x, logw=logw)
    ll = np.empty(1); ess = np.empty(1)
    ctx.call("smcn_normalise", _capi.dptr(ll), _capi.dptr(ess))
    wn, oll = orc.normalise_weights(logw)
    close(ll[0], oll, rtol=1e-14)
    close(ess[0], orc.calculate_ess(wn), rtol=1e-14)
    idx = ctx.resample(ll[0], np.log(N), 3, want_idx=True)
    u = orc.philox_particle_uniforms(9, 3, 0, N, 2, 0)
    np.testing.assert_array_equal(idx, orc.multinomial_indices(ctx.get_state(x=False, logw=False, wn=True)[2], u, "blocked"))
    xr = ctx.get_state()[0]
    np.testing.assert_array_equal(xr, x[idx])
    ctx.propose_nuts(0.01, 1.0, 5)
    r, xn, rn, _ = ctx.get_proposal()
    ref = orc.nuts_rvs(ot, xr, r, 1.0, 0.01, seed=9, iteration=5)
    np.testing.assert_array_equal(ctx.tree_stats()["nleap"], ref["nleap"])
    close(xn, ref["x_new"], rtol=1e-12, atol=1e-13)


def test_degenerate_weights_and_bad_particles():
    """-inf weights are masked out of the normalisation (samples.py:96-102); a particle
    whose density is not finite gets -inf (bridgestan.py:47-49), stops its tree at the
    first leaf and keeps its position; NaN weights poison the log-likelihood as in the
    reference."""
    from smcnuts_amd import ArmaModel, _capi
    t = ArmaModel()
    N = 512
    rng = np.random.default_rng(4)
    x = rng.normal(size=(N, 4)) * 0.05 + np.array([0, 0.9, 0, -1.8])
    x[3] = [0.0, 0.0, 0.0, 900.0]                 # sigma = exp(900): density -inf
    logw = rng.normal(size=N)
    logw[::7] = -np.inf
    ctx = _capi.Context(N, t.model_id, t.model_data)
    ctx.set_seed(1)
    ctx.set_state(x=x, logw=logw)
    ll = np.empty(1); ess = np.empty(1)
    ctx.call("smcn_normalise", _capi.dptr(ll), _capi.dptr(ess))
    wn, oll = orc.normalise_weights(logw)
    close(ll[0], oll, rtol=1e-14)
    got = ctx.get_state(x=False, logw=False, wn=True)[2]
    assert np.all(got[::7] == 0.0)
    close(got, wn, rtol=1e-14)
    ctx.propose_nuts(0.01, 1.0, 0)
    st = ctx.tree_stats()
    _, xn, _, _ = ctx.get_proposal()
    assert st["nleap"][3] == 1 and st["depth"][3] == 1
    np.testing.assert_array_equal(xn[3], x[3])
    lp0, ll0, lp1, ll1 = ctx.density_parts()
    assert not np.isfinite(lp0[3] + ll0[3])
    # all weights equal -> ESS = N exactly; all -inf -> loglik -inf
    ctx.set_state(logw=np.full(N, -3.25))
    ctx.call("smcn_normalise", _capi.dptr(ll), _capi.dptr(ess))
    close(ess[0], N, rtol=1e-14)
    close(ll[0], -3.25 + np.log(N), rtol=1e-14)
    lw = logw.copy(); lw[5] = np.nan
    ctx.set_state(logw=lw)
    ctx.call("smcn_normalise", _capi.dptr(ll), _capi.dptr(ess))
    assert np.isnan(ll[0])


def test_max_depth_zero_and_errors():
    """max_depth = 0: a single leapfrog per particle; API misuse fails loudly."""
    from smcnuts_amd import ArmaModel, SMCSampler, _capi
    t = ArmaModel()
    ctx = _capi.Context(64, t.model_id, t.model_data)
    ctx.set_seed(2)
    ctx.set_state(x=np.tile([0.0, 0.9, 0.0, -1.8], (64, 1)), logw=np.zeros(64))
    ctx.propose_nuts(0.01, 1.0, 0, max_depth=0)
    assert np.all(ctx.tree_stats()["nleap"] == 1)
    with pytest.raises(_capi.SmcnError):
        ctx.propose_nuts(0.01, 1.0, 0, max_depth=11)
    with pytest.raises(Exception, match="Unknown L-kernel supplied"):
        SMCSampler(K=1, N=64, target=t, step_size=0.01, lkernel="nonsense")
    with pytest.raises(TypeError):
        SMCSampler(K=1, N=64, target=object(), step_size=0.01)
    with pytest.raises(_capi.SmcnError):
        _capi.Context(16, 99, np.zeros(4))


@pytest.mark.parametrize("wide", [False, True])
def test_lane_queue_any_schedule_gives_the_same_run(wide):
    """The lane kernel's grid is capped (default: one wavefront per SIMD); with fewer lanes than particles every wavefront
    works through a contiguous run of particles, segment by segment (smcn_set_lane_grid, smcn_set_lane_segments).  Which
    lane runs a particle changes nothing about it: a ragged population under a wavefront per 64 particles (-1, the
    round-3 schedule), the default, and caps of 1 (more than 4 096 particles a wavefront: the plain launch), 2, 3 and 16
    wavefronts (every lane then runs up to 39 particles), with the particles' blocks worked off whole, in halves, thirds
    or single transitions, give the same fused run -- bit for bit with wide_eval=False; with the lane-group evaluation of
    stragglers on, the same trees and states to rounding."""
    from smcnuts_amd import ArmaModel, SMCSampler
    K, N, seed = 9, 5000, 4
    runs = {}
    # (cap, segments, longest block); segments: 0 = auto -- 4 (fewer where a wavefront's ready bits would not fit 64 words:
    # 2 500 particles a wavefront leave 1, 1 667 leave 2) --, 1 = whole blocks
    for cap, segs, fmax in ((-1, 0, 4), (0, 0, 4), (1, 0, 4), (2, 0, 4), (3, 0, 4), (16, 0, 4), (3, 1, 4), (3, 2, 4), (16, 3, 8),
                            (16, 8, 8), (40, 4, 8), (64, 0, 6)):
        s = SMCSampler(K=K, N=N, target=ArmaModel(), step_size=0.01, seed=seed, wide_eval=wide)
        s.samples.ctx.call("smcn_set_lane_grid", cap)
        s.samples.ctx.call("smcn_set_lane_segments", segs)
        s.run_fused(fuse_max=fmax)
        s.finalise_async()
        runs[(cap, segs, fmax)] = s
    ref = runs[(-1, 0, 4)]
    assert any(ref.resampled)
    for cap, s in runs.items():
        if wide:
            same_run_to_rounding(s, ref, f"lane grid {cap}")
        else:
            assert s.resampled == ref.resampled, cap
            np.testing.assert_array_equal(s.leapfrogs, ref.leapfrogs, err_msg=str(cap))
            np.testing.assert_array_equal(s.x_saved, ref.x_saved, err_msg=str(cap))
            np.testing.assert_array_equal(s.logw_saved, ref.logw_saved, err_msg=str(cap))
            np.testing.assert_array_equal(s.ess, ref.ess, err_msg=str(cap))
            np.testing.assert_array_equal(s.mean_estimate, ref.mean_estimate, err_msg=str(cap))


def test_lane_schedules_drawn_at_random_give_the_same_run():
    """The schedule of the lane kernel -- wavefronts launched, segments a block is worked off in, longest block -- drawn at
    random for three ragged populations: particles per wavefront from 65 to 2 600, 1 to 8 segments (more than a block has
    transitions included), blocks of 2 to 12 iterations.  With wide_eval=False every one of them is the run of the plain
    schedule (a wavefront per 64 particles), bit for bit."""
    from smcnuts_amd import ArmaModel, SMCSampler
    rng = np.random.default_rng(2026)

    def run(N, K, cap, segs, fmax):
        s = SMCSampler(K=K, N=N, target=ArmaModel(), step_size=0.01, seed=9, wide_eval=False)
        s.samples.ctx.call("smcn_set_lane_grid", cap)
        s.samples.ctx.call("smcn_set_lane_segments", segs)
        s.run_fused(fuse_max=fmax)
        s.finalise_async()
        out = (list(s.resampled), s.leapfrogs.copy(), s.x_saved.copy(), s.logw_saved.copy(), s.ess.copy())
        s.samples.ctx.close()
        return out

    for N, K in ((1301, 7), (2600, 6), (7777, 5)):
        ref = run(N, K, -1, 0, 4)
        assert any(ref[0])
        for _ in range(4):
            cap = int(rng.integers(1, max(2, N // 65)))
            segs = int(rng.integers(0, 9))
            fmax = int(rng.integers(2, 13))
            got = run(N, K, cap, segs, fmax)
            what = f"N={N} K={K} cap={cap} segments={segs} fuse_max={fmax}"
            assert got[0] == ref[0], what
            for a, b in zip(got[1:], ref[1:]):
                np.testing.assert_array_equal(a, b, err_msg=what)
    # ... and at the sizes the scheduler is for: ragged populations beyond one wavefront per SIMD under the default grid
    for N, K, fmax in ((100003, 5, 8), (200017, 4, 4)):
        ref = run(N, K, -1, 0, fmax)
        got = run(N, K, 0, 0, fmax)
        assert got[0] == ref[0]
        for a, b in zip(got[1:], ref[1:]):
            np.testing.assert_array_equal(a, b, err_msg=f"N={N}")


@pytest.mark.parametrize("name", ["arma_fwd"])
def test_lane_queue_on_reference_tapes(golden_dir, name):
    """ONE wavefront for the reference's 128 particles: every lane runs two particles, the second a ready job of its
    wavefront -- draws consumed and states are the reference's recorded ones."""
    from smcnuts_amd import _capi
    g = load(golden_dir, name)
    t, _ = targets(name)
    N = int(g["N"])
    assert N > 64
    ctx = _capi.Context(N, t.model_id, t.model_data)
    ctx.call("smcn_set_lane_grid", 1)
    for k in range(int(g["K"])):
        ctx.set_state(x=g[f"x_in_{k}"], logw=np.zeros(N))
        ctx.call("smcn_set_momentum", _capi.dptr(np.ascontiguousarray(g[f"r_{k}"])))
        ctx.propose_nuts(float(g["eps"]), float(g[f"phi_prop_{k}"]), k, tape=g[f"tape_{k}"], tape_off=g[f"tape_off_{k}"])
        _, xn, rn, _ = ctx.get_proposal(r=False)
        st = ctx.tree_stats()
        assert not st["flags"].any()
        np.testing.assert_array_equal(st["ndraws"], np.diff(g[f"tape_off_{k}"]))
        close(xn, g[f"x_new_{k}"], rtol=1e-13, atol=1e-14)
        close(rn, g[f"r_new_{k}"], rtol=2e-12, atol=2e-13)
