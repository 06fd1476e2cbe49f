"""Target densities of the oracle (parity UNPINNED against BridgeStan, which is
absent): independent NumPy/SciPy restatement of the .stan programs, finite
differences, and the posterior means of stan_models/<m>/<m>.params."""
import json
import os

import numpy as np
import pytest
from scipy.signal import lfilter
from scipy.special import gammaln

from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "smcnuts_amd", "model", "data")


def arma_np(x, y, phi):
    """arma.stan:14-30 via an IIR filter (err_t = c_t - theta err_{t-1})."""
    mu, beta, theta, s = x
    sigma = np.exp(s)
    c = np.empty_like(y)
    c[0] = y[0] - (mu + beta * mu)
    c[1:] = y[1:] - mu - beta * y[:-1]
    err = lfilter([1.0], [1.0, theta], c)
    norm = lambda v, sd: -0.5 * np.log(2 * np.pi) - np.log(sd) - 0.5 * (v / sd) ** 2
    lpri = norm(mu, 10) + norm(beta, 2) + norm(theta, 2) - np.log(np.pi * 2.5 * (1 + (sigma / 2.5) ** 2)) + s
    llik = np.sum(norm(err, sigma))
    return lpri + phi * llik


def prmwcd_np(x, d, phi):
    """PRMwCD.stan:17-38."""
    M, C, q = d["M"], d["Clength"], d["q"]
    y = np.asarray(d["y"], float)
    X = np.asarray(d["Xkernel"]).reshape(d["N"], C)
    beta, g = x[:M], x[M]
    Gam = np.exp(g)
    lp = 2 * np.log(1.3) - gammaln(2.0) - 3 * np.log(Gam) - 1.3 / Gam + g
    lp += np.sum(-np.log(Gam) - np.abs(beta[1:] / Gam) ** q)
    eta = beta[0] + X @ beta[1:]
    ll = np.sum(y * eta - np.exp(eta) - gammaln(y + 1))
    return lp + phi * ll


def fd_grad(f, x, h=1e-6):
    g = np.zeros_like(x)
    for i in range(x.size):
        e = np.zeros_like(x); e[i] = h
        g[i] = (f(x + e) - f(x - e)) / (2 * h)
    return g


@pytest.mark.parametrize("phi", [1.0, 0.3, 0.0])
def test_arma_density_and_gradient(phi):
    d = json.load(open(os.path.join(DATA, "arma.json")))
    y = np.asarray(d["y"])
    t = orc.OracleTarget(orc.MODEL_ARMA, orc.arma_data(os.path.join(DATA, "arma.json")), 4)
    rng = np.random.default_rng(0)
    for _ in range(20):
        x = rng.normal(size=4) * np.array([0.3, 0.5, 0.5, 0.7]) + np.array([0, 0.5, 0, -1.0])
        ref = arma_np(x, y, phi)
        np.testing.assert_allclose(t.logpdf(x, phi), ref, rtol=1e-11, atol=1e-9)
        g = t.logpdfgrad(x, phi)
        np.testing.assert_allclose(g, fd_grad(lambda z: arma_np(z, y, phi), x), rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize("phi", [1.0, 0.25])
def test_prmwcd_density_and_gradient(phi):
    d = json.load(open(os.path.join(DATA, "PRMwCD.json")))
    t = orc.OracleTarget(orc.MODEL_PRMWCD, orc.prmwcd_data(os.path.join(DATA, "PRMwCD.json")), 13)
    rng = np.random.default_rng(1)
    for _ in range(20):
        x = rng.normal(size=13) * 0.7
        ref = prmwcd_np(x, d, phi)
        np.testing.assert_allclose(t.logpdf(x, phi), ref, rtol=1e-11, atol=1e-9)
        np.testing.assert_allclose(t.logpdfgrad(x, phi), fd_grad(lambda z: prmwcd_np(z, d, phi), x),
                                   rtol=2e-5, atol=2e-4)


def test_failure_convention():
    """bridgestan.py:45-49,77-80: anything non-finite -> -inf / grad of -inf."""
    t = orc.OracleTarget(orc.MODEL_ARMA, orc.arma_data(os.path.join(DATA, "arma.json")), 4)
    x = np.array([0.0, 0.0, 0.0, 800.0])     # sigma = exp(800) = inf
    assert t.logpdf(x) == -np.inf
    assert np.all(np.isneginf(t.logpdfgrad(x)))
    assert t.logpdf(np.array([np.nan, 0, 0, 0.0])) == -np.inf


def test_arma_posterior_means_match_params_file():
    """stan_models/arma/arma.params:1-4, column 2 (posterior means of mu, beta
    [named "phi" there], theta, sigma).  Philox-mode oracle SMC, N=1024."""
    truth = np.array([0.00678443422162953, 0.9570083053800078, -0.03407898212798232, 0.1666098193000008])
    t = orc.OracleTarget(orc.MODEL_ARMA, orc.arma_data(os.path.join(DATA, "arma.json")), 4)
    N, K = 1024, 25
    x0 = orc.philox_normals(7, 0, N, 4, 3)
    logq0 = orc.std_normal_logpdf(x0)
    out = orc.smc_run(t, K, N, 0.01, x0, logq0, seed=7)
    est = out["mean_estimate"][-1]
    # posterior sds are ~(0.011, 0.023, 0.059, 0.008) (column 3 of the file)
    assert np.all(np.abs(est - truth) < np.array([0.004, 0.008, 0.02, 0.003])), est


def test_oracle_c_under_address_and_ub_sanitizers(tmp_path):
    """The C restatement built with -fsanitize=address,undefined replays a golden NUTS transition (deep trees:
    every stack level of the recursion) and a Philox run in a child process; any report fails the test
    (the sanitizer build is the CPU side's race / memory checker)."""
    import subprocess
    import sys
    so = str(tmp_path / "liboracle_asan.so")
    r = subprocess.run(["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=undefined", "-ffp-contract=off", "-fPIC", "-shared", "-o", so,
                        orc.SRC_PATH, "-lm"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime on this host: " + r.stderr[-300:])
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    code = f"""
import sys, numpy as np
sys.path.insert(0, {ROOT!r})
from oracle import oracle as orc
v = orc.load_variant({so!r})
g = np.load({os.path.join(ROOT, "tests", "golden", "gauss4_deep.npz")!r})
t = orc.OracleTarget(orc.MODEL_GAUSS, orc.gauss_data(4), 4)
res = orc.nuts_rvs(t, g["x_in_0"], g["r_0"], float(g["phi_prop_0"]), float(g["eps"]), tape=g["tape_0"], tape_off=g["tape_off_0"], clib=v)
assert (res["ndraws"] == np.diff(g["tape_off_0"])).all()
a = orc.OracleTarget(orc.MODEL_ARMA, orc.arma_data({os.path.join(DATA, "arma.json")!r}), 4)
x = np.random.default_rng(0).normal(size=(64, 4)) * 0.05 + np.array([0, 0.9, 0, -1.8])
r = orc.nuts_rvs(a, x, np.random.default_rng(1).normal(size=(64, 4)), 1.0, 0.01, seed=3, iteration=1, clib=v)
assert r["nleap"].sum() > 64
print("SAN-OK")
"""
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "SAN-OK" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
