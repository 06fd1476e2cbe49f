"""Particle sharding over GPUs: one process per GPU, the only traffic is an
all-gather of a few doubles per reduction; every rank then
combines the shard partials in rank order, so all ranks hold bit-identical
scalars."""
import numpy as np


class SingleProcess:
    rank, world_size = 0, 1

    def allgather(self, v):
        return np.asarray(v, dtype=np.float64)[None, :]


class TorchDistComm:
    """torch.distributed plumbing ("nccl" = RCCL over xGMI on the GPU box,
    "gloo" in the CPU tests)."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._torch, self._dist = torch, dist
        self.rank, self.world_size = dist.get_rank(), dist.get_world_size()
        self.device = device if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu"))

    def allgather(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        t = self._torch.as_tensor(v.reshape(-1)).to(self.device)
        out = self._torch.empty(self.world_size * t.numel(), dtype=t.dtype, device=self.device)
        self._dist.all_gather_into_tensor(out, t)
        return out.cpu().numpy().reshape((self.world_size,) + v.shape)


def combine_lse_partials(parts):
    """Combine per-shard [max, count, s1, s2] (smcn_normalise_partials) in rank
    order into the global (loglik, sum wn^2), following scipy.special.logsumexp
    (max elements taken out of the sum; log1p(s/m) + log m + max) as
    Samples.normalise_weights does (samples/samples.py:96-105)."""
    parts = np.asarray(parts, dtype=np.float64).reshape(-1, 4)
    mx = parts[:, 0]
    if np.any(np.isnan(mx)):
        return np.nan, np.nan
    M = np.max(mx)
    if M == -np.inf:      # every weight is -inf: logsumexp of an empty selection
        return -np.inf, np.nan
    shift = M if np.isfinite(M) else 0.0
    m = s = s2 = 0.0
    with np.errstate(all="ignore"):
        for g in range(parts.shape[0]):
            mg, cnt, s1g, s2g = parts[g]
            if mg == -np.inf:
                continue
            sg = mg if np.isfinite(mg) else 0.0
            scale = np.exp(sg - shift)
            if mg == M:
                m += cnt
                s += s1g * scale
            else:
                s += (s1g + cnt) * scale
            s2 += s2g * scale * scale
        sm = s if s == 0 else s / m
        ll = np.log1p(sm) + np.log(m) + M
        sum_wn2 = s2 * np.exp(2.0 * (shift - ll))
    return float(ll), float(sum_wn2)
