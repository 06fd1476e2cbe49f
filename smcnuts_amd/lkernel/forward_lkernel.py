"""Forward-proposal L-kernel.  Mirror of smcnuts/lkernel/forward_lkernel.py:4-35."""
import numpy as np

from .. import _capi


class ForwardLKernel:
    def __init__(self, target, momentum_proposal):
        self.target = target
        self.momentum_proposal = momentum_proposal

    def calculate_L(self, r_new, _):
        """forward_lkernel.py:22-35 on host arrays (plug-in interface)."""
        return self.momentum_proposal.logpdf(np.multiply(-1, r_new))

    def apply(self, ctx, forward_kernel):
        """Device path: with the N(0, I) momentum proposal L and q are closed
        forms inside the re-weight kernel; a duck-typed proposal is evaluated by
        the caller's own object and handed over."""
        if not forward_kernel.native_momentum:
            r, _, r_new, _ = ctx.get_proposal(x_new=False)
            L = np.ascontiguousarray(self.calculate_L(r_new, None), dtype=np.float64)
            q = np.ascontiguousarray(forward_kernel.logpdf(r), dtype=np.float64)
            ctx.call("smcn_set_lkernel_values", _capi.dptr(L), _capi.dptr(q))
        return _capi.LKERNEL_FORWARD
