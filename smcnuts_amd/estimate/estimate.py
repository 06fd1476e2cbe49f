"""Weighted mean / variance in constrained space.
Mirror of smcnuts/estimate/estimate.py:4-95 (`Estimate`)."""
import numpy as np


class Estimate:
    def __init__(self, target):
        self.target = target

    def return_estimate_device(self, ctx, comm=None):
        """estimate.py:38-57,79-95 on the resident shard: two passes as the
        reference (mean, then weighted squared deviations from that mean)."""
        def total(s):
            return comm.allgather(s).sum(axis=0) if comm is not None and comm.world_size > 1 else s
        mean = total(ctx.moment_sums(None))
        var = total(ctx.moment_sums(mean))
        return mean, var

    def return_estimate(self, x, wn):
        """The reference's plug-in signature on host arrays."""
        _x = self.target.constrain(x) if hasattr(self.target, "constrained_dim") else np.array(x, copy=True)
        mean = wn.T @ _x
        var = wn.T @ np.square(_x - mean)
        return mean, var
