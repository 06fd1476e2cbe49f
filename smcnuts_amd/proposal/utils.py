"""Host-array form of the Metropolis test of the asymptotic strategy
(reference: smcnuts/proposal/utils.py:3-34, `hmc_accept_reject`); the sampler itself
runs smcn_accept_reject on the device."""
import numpy as np


def hmc_accept_mask(logp_old, logp_new, r_old, r_new, x_new, u):
    """Vectorised over particles: True where the move is kept.  Rejected iff
    u > min(1, exp(H_new - H_old)) or the proposal has an infinite coordinate;
    a NaN ratio accepts (Python's min(1., nan) is 1.)."""
    r_old, r_new, x_new = (np.atleast_2d(a) for a in (r_old, r_new, x_new))
    with np.errstate(all="ignore"):
        dH = (np.asarray(logp_new) - 0.5 * np.einsum("ij,ij->i", r_new, r_new)) \
           - (np.asarray(logp_old) - 0.5 * np.einsum("ij,ij->i", r_old, r_old))
        ratio = np.exp(dH)
    prob = np.where(ratio < 1.0, ratio, 1.0)
    return ~((np.asarray(u) > prob) | np.isinf(x_new).any(axis=1))


def hmc_accept_reject(target_lpdf, x, x_prime, r, r_prime, phi=1.0, rng=None):
    """Single-particle plug-in with the reference's call shape."""
    rng = np.random.default_rng() if rng is None else rng
    keep = hmc_accept_mask(target_lpdf(x, phi=phi), target_lpdf(x_prime, phi=phi), r, r_prime, x_prime, rng.uniform())
    return bool(keep[0])
