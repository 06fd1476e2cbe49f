"""NUTS followed by a Metropolis accept/reject per particle (the forward kernel of
the asymptotic L-kernel strategy).  Mirror of smcnuts/proposal/nuts_acc_rej.py:8-52."""
import numpy as np

from .. import _capi
from .nuts import NUTSProposal


class NUTSProposalWithAccRej(NUTSProposal):
    def propose(self, ctx, phi, iteration, tape=None, tape_off=None, r=None, u_accept=None):
        super().propose(ctx, phi, iteration, tape=tape, tape_off=tape_off, r=r)
        u = None if u_accept is None else np.ascontiguousarray(u_accept, dtype=np.float64)
        ctx.call("smcn_accept_reject", float(phi), _capi.dptr(u), int(iteration))     # nuts_acc_rej.py:44-49

    def rvs(self, x_cond, r_cond, phi=1.0, tape=None, tape_off=None, seed=None, u_accept=None):
        super().rvs(x_cond, r_cond, phi, tape=tape, tape_off=tape_off, seed=seed)
        c = self._ctx
        u = None if u_accept is None else np.ascontiguousarray(u_accept, dtype=np.float64)
        c.call("smcn_accept_reject", float(phi), _capi.dptr(u), int(self._calls - 1))
        _, x_new, r_new, _ = c.get_proposal(r=False)
        return x_new, r_new
