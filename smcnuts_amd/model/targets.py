"""Device-native targets with the reference's StanModel surface.

Mirror of smcnuts/model/bridgestan.py:7-146 (`StanModel`): `.dim`,
`.constrained_dim`, `.param_names`, `.logpdf(x, phi=1.0)`,
`.logpdfgrad(x, phi=1.0)`, `.constrain(x)`.  The reference's back end is
BridgeStan (host-only, one compiled Stan model per .so); here each model is a
device functor restated from its .stan text (smcnuts_amd/csrc/smcn_models.hpp)
and the temperature phi is a kernel argument (the reference rewrites the JSON
data file and reloads the model on every change, bridgestan.py:122-146).
"""
import json
import os

import numpy as np

from .. import _capi

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


class DeviceTarget:
    """Base: a model id + flat fp64 data block understood by the HIP library."""

    model_id = None

    def __init__(self, model_data, dim, param_names):
        self.model_data = np.ascontiguousarray(model_data, dtype=np.float64)
        self.dim = int(dim)
        self.constrained_dim = int(dim)
        self._param_names = list(param_names)
        self._ctx = None     # small private context for the host-facing batched calls
        self.device = 0

    def param_names(self):
        return list(self._param_names)

    def _context(self, M):
        if self._ctx is None or self._ctx.N < M:
            if self._ctx is not None:
                self._ctx.close()
            self._ctx = _capi.Context(max(int(M), 256), self.model_id, self.model_data, device=self.device)
        return self._ctx

    # bridgestan.py:28-58: 1-D -> scalar, 2-D -> [N]; failures -> -inf
    def logpdf(self, x, phi=1.0, adjust_transform=True):
        x = np.asarray(x, dtype=np.float64)
        x2 = np.atleast_2d(x)
        lp = self._context(x2.shape[0]).target_eval(x2, phi)[0]
        return float(lp[0]) if x.ndim == 1 else lp

    # bridgestan.py:60-90
    def logpdfgrad(self, x, phi=1.0, adjust_transform=True):
        x = np.asarray(x, dtype=np.float64)
        x2 = np.atleast_2d(x)
        g = self._context(x2.shape[0]).target_eval(x2, phi, want_grad=True)[1]
        return g[0] if x.ndim == 1 else g

    def logpdf_parts(self, x):
        """(log prior incl. Jacobian, log likelihood); log pi_phi = lpri + phi*llik."""
        x2 = np.atleast_2d(np.asarray(x, dtype=np.float64))
        _, _, a, b = self._context(x2.shape[0]).target_eval(x2, 1.0, want_parts=True)
        return a, b

    # bridgestan.py:93-120
    def constrain(self, x, include_tparams=True, include_gqs=True):
        x = np.asarray(x, dtype=np.float64)
        x2 = np.atleast_2d(x)
        c = self._context(x2.shape[0]).constrain(x2)
        return c[0] if x.ndim == 1 else c


class GaussianTarget(DeviceTarget):
    """prior N(0, prior_sd^2 I) x optional likelihood N(x | lik_mean 1, lik_sd^2 I)."""
    model_id = _capi.MODEL_GAUSS

    def __init__(self, dim, prior_sd=1.0, lik_mean=None, lik_sd=1.0):
        has = 0.0 if lik_mean is None else 1.0
        data = [dim, prior_sd, has, 0.0 if lik_mean is None else lik_mean, lik_sd]
        super().__init__(data, dim, [f"x.{i + 1}" for i in range(dim)])


class IsoGaussian(GaussianTarget):
    """log pi = -|x|^2/2 - D/2 log 2 pi (BASELINE config 5)."""

    def __init__(self, dim):
        super().__init__(dim)


class HostTarget:
    """Adapter for targets evaluated on the HOST: any object with the reference's StanModel surface
    (smcnuts/model/bridgestan.py:7-146: `.dim`, `.logpdf(x, phi)`, `.logpdfgrad(x, phi)`, optional
    `.constrain(x)` / `.constrained_dim` / `.param_names()`), e.g. a BridgeStan model that has no device
    functor here.  The NUTS tree building, the weights, the resampling and the estimates still run on
    the GPU; the library calls back for the density (`smcn_set_host_target`), in lock step for all
    particles -- one call per leapfrog of the longest tree: the generality path, not the fast one.

    The callback hands the library log prior and log likelihood separately, taken from the wrapped
    model the way the reference's tempering does (adaptive_tempering.py:44-49):
    lpri = logpdf(x, phi=0), llik = logpdf(x, phi=1) - lpri (the same for the gradients)."""
    model_id = _capi.MODEL_HOST
    host_evaluated = True

    def __init__(self, target):
        self.target = target
        self.dim = int(target.dim)
        self.constrained_dim = int(getattr(target, "constrained_dim", self.dim))
        self.model_data = np.array([float(self.dim)])
        self.device = 0
        self.calls = 0
        self._keep = []          # the ctypes trampolines must outlive the contexts they are registered with

    def param_names(self):
        f = getattr(self.target, "param_names", None)
        return list(f()) if callable(f) else [f"x.{i + 1}" for i in range(self.dim)]

    def logpdf(self, x, phi=1.0, **kw):
        return self.target.logpdf(x, phi=phi)

    def logpdfgrad(self, x, phi=1.0, **kw):
        return self.target.logpdfgrad(x, phi=phi)

    def logpdf_parts(self, x):
        with np.errstate(all="ignore"):
            a = np.asarray(self.target.logpdf(x, phi=0.0), dtype=np.float64)
            return a, np.asarray(self.target.logpdf(x, phi=1.0), dtype=np.float64) - a

    def constrain(self, x, **kw):
        f = getattr(self.target, "constrain", None)
        return f(x) if callable(f) else np.array(x, copy=True)

    def attach(self, ctx):
        """Register the density callback with a context created for this target."""
        import ctypes as C
        D = self.dim

        def trampoline(user, n, d, x, want_grad, lpri, llik, gpri, glik):
            try:
                xs = np.ctypeslib.as_array(x, shape=(n, d))
                with np.errstate(all="ignore"):
                    a = np.asarray(self.target.logpdf(xs, phi=0.0), dtype=np.float64).reshape(n)
                    b = np.asarray(self.target.logpdf(xs, phi=1.0), dtype=np.float64).reshape(n)
                    np.ctypeslib.as_array(lpri, shape=(n,))[:] = a
                    np.ctypeslib.as_array(llik, shape=(n,))[:] = b - a
                    if want_grad:
                        ga = np.asarray(self.target.logpdfgrad(xs, phi=0.0), dtype=np.float64).reshape(n, d)
                        gb = np.asarray(self.target.logpdfgrad(xs, phi=1.0), dtype=np.float64).reshape(n, d)
                        np.ctypeslib.as_array(gpri, shape=(n, d))[:] = ga
                        np.ctypeslib.as_array(glik, shape=(n, d))[:] = gb - ga
                self.calls += 1
                return 0
            except Exception as e:      # the library turns this into an error of the calling entry point
                self.last_error = e
                return 1

        fn = _capi.HOST_TARGET_FN(trampoline)
        self._keep.append(fn)
        ctx.call("smcn_set_host_target", fn, None)
        return ctx


def as_target(target):
    """Device-native targets pass through; anything else with the StanModel surface is wrapped."""
    if hasattr(target, "model_id"):
        return target
    for name in ("dim", "logpdf", "logpdfgrad"):
        if not hasattr(target, name):
            raise TypeError(f"target needs .{name} (the reference's StanModel interface, model/bridgestan.py)")
    return HostTarget(target)


def _load_json(path):
    s = open(path).read().rstrip()
    if s.endswith('"phi":'):       # the shipped PRMwCD.json is truncated
        s += " 1.0}"
    return json.loads(s)


class ArmaModel(DeviceTarget):
    """stan_models/arma/arma.stan; unconstrained (mu, beta, theta, log sigma)."""
    model_id = _capi.MODEL_ARMA

    def __init__(self, data_path=None):
        d = _load_json(data_path or os.path.join(DATA_DIR, "arma.json"))
        y = np.asarray(d["y"], dtype=np.float64)
        if int(d["T"]) != y.size:
            raise ValueError("arma data: T != len(y)")
        super().__init__(np.concatenate([[float(y.size)], y]), 4, ["mu", "beta", "theta", "sigma"])


class PRMwCDModel(DeviceTarget):
    """stan_models/PRMwCD/PRMwCD.stan; unconstrained (Beta[1..M], log Gamma)."""
    model_id = _capi.MODEL_PRMWCD

    def __init__(self, data_path=None):
        d = _load_json(data_path or os.path.join(DATA_DIR, "PRMwCD.json"))
        M = int(d["M"])
        data = np.concatenate([[float(d["N"]), float(M), float(d["Clength"]), float(d["q"])],
                               np.asarray(d["y"], dtype=np.float64), np.asarray(d["Xkernel"], dtype=np.float64)])
        if M != int(d["Clength"]) + 1:
            raise ValueError("PRMwCD data: M must equal Clength + 1")
        super().__init__(data, M + 1, [f"Beta.{i + 1}" for i in range(M)] + ["Gamma"])
        # the data shape the device functors are specialised for (unrolled observation loop, wave-per-tree finisher):
        # trees that want more than 9 doublings are then parked and finished one per wavefront (Samples: nuts_cap="auto")
        self.two_phase_default = (9, True, 8) if (96 < int(d["N"]) <= 100 and int(d["Clength"]) == 11 and float(d["q"]) == 0.5) else None


def StanModel(model_name, model_path=None, data_path=None):
    """Same call shape as the reference's StanModel(model_name, model_path,
    data_path) (bridgestan.py:13); resolves the model NAME to its device
    functor -- arbitrary .stan files are not compiled here."""
    name = str(model_name).lower()
    if name == "arma":
        return ArmaModel(data_path)
    if name == "prmwcd":
        return PRMwCDModel(data_path)
    raise NotImplementedError(
        f"no device-native functor for Stan model {model_name!r} (available: arma, PRMwCD).  Pass the model "
        "object itself -- anything with .dim / .logpdf(x, phi) / .logpdfgrad(x, phi), e.g. the reference's "
        "StanModel over BridgeStan -- as `target`: it is evaluated on the host through HostTarget.")
