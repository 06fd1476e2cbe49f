"""Weighted mean / variance in constrained space.
Mirror of smcnuts/estimate/estimate.py:4-95 (`Estimate`)."""
import numpy as np

from .._capi import dptr as _dptr


def device_moments(target, ctx, total=lambda s: s):
    """mean = sum wn c(x), var = sum wn (c(x) - mean)^2 of the resident particles (estimate.py:79-95)."""
    if getattr(target, "host_evaluated", False) and callable(getattr(target.target, "constrain", None)):
        # the caller's own constrain() (bridgestan.py:100-120) on the host, the weighted sums on the device
        xc = np.ascontiguousarray(target.constrain(ctx.get_state(logw=False)[0]), dtype=np.float64)
        Dc = xc.shape[1]
        mean, var = np.empty(Dc), np.empty(Dc)
        ctx.call("smcn_moment_sums_of", _dptr(xc), Dc, None, _dptr(mean))
        mean = np.ascontiguousarray(total(mean))
        ctx.call("smcn_moment_sums_of", _dptr(xc), Dc, _dptr(mean), _dptr(var))
        return mean, total(var)
    mean = total(ctx.moment_sums(None))
    var = total(ctx.moment_sums(mean))
    return mean, var


class Estimate:
    def __init__(self, target):
        self.target = target

    def return_estimate_device(self, ctx, comm=None):
        """estimate.py:38-57,79-95 on the resident shard: two passes as the
        reference (mean, then weighted squared deviations from that mean)."""
        def total(s):
            return comm.allgather(s).sum(axis=0) if comm is not None and comm.world_size > 1 else s
        return device_moments(self.target, ctx, total)

    def return_estimate(self, x, wn):
        """The reference's plug-in signature on host arrays."""
        _x = self.target.constrain(x) if hasattr(self.target, "constrained_dim") else np.array(x, copy=True)
        mean = wn.T @ _x
        var = wn.T @ np.square(_x - mean)
        return mean, var
