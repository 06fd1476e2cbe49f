"""oracle/pynuts.py -- TEST INFRASTRUCTURE ONLY.  NOT PRODUCT CODE.

The reference-SHAPED CPU path: NUTSProposal (smcnuts/proposal/nuts.py:6-189) restated in
plain Python the way the reference runs it -- a serial `for i in range(N)` over particles
(nuts.py:50), a recursive build_tree (nuts.py:114-150), one leapfrog per leaf
(nuts.py:162-175) -- with the target reached through ctypes, two calls per leapfrog
(`logpdfgrad` nuts.py:171, then `logpdf` nuts.py:122), exactly where the reference crosses
into BridgeStan (model/bridgestan.py:46,78).  BridgeStan is absent here, so the density
behind the boundary is the C restatement of the .stan program (oracle/smcnuts_oracle.c).

Used (a) by tests/test_oracle_golden.py, which replays the draws recorded from the real
reference and requires identical draw counts and x', r' -- so this file is pinned the same
way oracle/smcnuts_oracle.c is -- and (b) by bench.py's cpu_baseline leg as the
"Python-serial, one core" timing that SURVEY.md 8(d) asks for beside the C port.
"""
import numpy as np

MAX_TREE_DEPTH = 10          # nuts.py:4


class TapeRNG:
    """Replays one particle's recorded draws in consumption order."""

    def __init__(self, tape):
        self.t, self.i = tape, 0

    def _next(self):
        v = self.t[self.i]
        self.i += 1
        return v

    def exponential(self, scale=1.0):
        return self._next()          # the tape holds the Exp(1) value itself (make_golden.py)

    def uniform(self, lo=0.0, hi=1.0):
        return self._next()


class PyNUTS:
    def __init__(self, target, step_size, rng=None):
        self.target, self.step_size, self.rng = target, step_size, rng
        self.nleap = 0

    # nuts.py:34-56
    def rvs(self, x_cond, r_cond, phi, tapes=None):
        x_prime, r_prime = np.zeros_like(x_cond), np.zeros_like(r_cond)
        self.ndraws = np.zeros(len(x_cond), dtype=np.int64)
        for i in range(len(x_cond)):
            if tapes is not None:
                self.rng = TapeRNG(tapes[i])
            x_prime[i], r_prime[i] = self.generate_nuts_samples(x_cond[i], r_cond[i], phi)
            if tapes is not None:
                self.ndraws[i] = self.rng.i
        return x_prime, r_prime

    # nuts.py:58-112
    def generate_nuts_samples(self, x0, r0, phi):
        t, rng = self.target, self.rng
        logp = t.logpdf(x0, phi)
        H0 = logp - 0.5 * np.dot(r0, r0.T)
        logu = float(H0 - rng.exponential(1))
        grad = t.logpdfgrad(x0, phi)
        x, r = x0, r0
        xm, xp, rm, rp, gm, gp = x0, x0, r0, r0, grad, grad
        depth, n, stop = 0, 1, 0
        while stop == 0:
            direction = int(2 * (rng.uniform(0, 1) < 0.5) - 1)
            if direction == -1:
                xm, rm, gm, _, _, _, xq, rq, nq, sq = self.build_tree(xm, rm, gm, logu, direction, depth, phi)
            else:
                _, _, _, xp, rp, gp, xq, rq, nq, sq = self.build_tree(xp, rp, gp, logu, direction, depth, phi)
            if sq == 0 and rng.uniform() < min(1.0, float(nq) / float(n)):     # nuts.py:99 (short-circuit)
                x, r = xq, rq
            n += nq
            stop = sq or self.stop_criterion(xm, xp, rm, rp)
            depth += 1
            if depth > MAX_TREE_DEPTH:
                break
        return x, r

    # nuts.py:114-150
    def build_tree(self, x, r, grad, logu, direction, depth, phi):
        if depth == 0:
            xq, rq, gq = self.leapfrog(x, r, grad, direction, phi)
            logpq = self.target.logpdf(xq, phi)
            joint = logpq - 0.5 * np.dot(rq, rq)
            nq = int(logu < joint)
            sq = int((logu - 100.0) >= joint)
            return xq, rq, gq, xq, rq, gq, xq, rq, nq, sq
        xm, rm, gm, xp, rp, gp, xq, rq, nq, sq = self.build_tree(x, r, grad, logu, direction, depth - 1, phi)
        if sq == 0:
            if direction == -1:
                xm, rm, gm, _, _, _, xq2, rq2, nq2, sq2 = self.build_tree(xm, rm, gm, logu, direction, depth - 1, phi)
            else:
                _, _, _, xp, rp, gp, xq2, rq2, nq2, sq2 = self.build_tree(xp, rp, gp, logu, direction, depth - 1, phi)
            if self.rng.uniform() < (float(nq2) / max(float(int(nq) + int(nq2)), 1.0)):
                xq, rq = xq2, rq2
            nq = int(nq) + int(nq2)
            sq = int(sq or sq2 or self.stop_criterion(xm, xp, rm, rp))
        return xm, rm, gm, xp, rp, gp, xq, rq, nq, sq

    # nuts.py:152-160
    @staticmethod
    def stop_criterion(xm, xp, rm, rp):
        dx = xp - xm
        return (np.dot(dx, rm.T) < 0) or (np.dot(dx, rp.T) < 0)

    # nuts.py:162-175
    def leapfrog(self, x, r, grad, direction, phi):
        eps = self.step_size
        rq = np.add(r, (direction * eps / 2) * grad)
        xq = np.add(x, direction * eps * rq)
        gq = self.target.logpdfgrad(xq, phi)
        rq = np.add(rq, (direction * eps / 2) * gq)
        self.nleap += 1
        return xq, rq, gq
