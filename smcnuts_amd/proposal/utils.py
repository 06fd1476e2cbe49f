"""Mirror of smcnuts/proposal/utils.py:3-34 (`hmc_accept_reject`) on host arrays."""
import numpy as np


def hmc_accept_reject(target_lpdf, x, x_prime, r, r_prime, phi=1.0, rng=np.random.default_rng()):
    """True if the move is accepted (plug-in interface; the sampler runs
    smcn_accept_reject on the device)."""
    with np.errstate(all="ignore"):
        H1 = target_lpdf(x_prime, phi=phi) - (0.5 * np.dot(r_prime, r_prime))
        H0 = target_lpdf(x, phi=phi) - (0.5 * np.dot(r, r))
        acceptance_probability = min(1., np.exp(H1 - H0))
        if rng.uniform() > acceptance_probability or np.any(np.isinf(x_prime)):
            return False
        return True
