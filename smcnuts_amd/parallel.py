"""Particle sharding over GPUs: one process per GPU, the only traffic is an
all-gather of a few doubles per reduction; every rank then
combines the shard partials in rank order, so all ranks hold bit-identical
scalars."""
import os

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
        self.device_path = False
        if self.device.type == "cuda" and os.environ.get("SMCN_HOST_EXCHANGE", "0") != "1":
            self.device_path = self._self_test()

    def _self_test(self):
        """All-gather through aliased raw device pointers once and check it; any
        failure falls back to the host exchange."""
        try:
            t = self._torch
            n = 6
            src = t.arange(n, dtype=t.float64, device=self.device) + 100.0 * self.rank
            dst = t.zeros(n * self.world_size, dtype=t.float64, device=self.device)
            self.allgather_device(src.data_ptr(), dst.data_ptr(), n)
            t.cuda.synchronize(self.device)
            want = t.cat([t.arange(n, dtype=t.float64) + 100.0 * r for r in range(self.world_size)])
            ok = bool(t.equal(dst.cpu(), want))
        except Exception:
            ok = False
        flag = t.tensor([1.0 if ok else 0.0], device=self.device)
        self._dist.all_reduce(flag, op=self._dist.ReduceOp.MIN)
        return bool(flag.item() == 1.0)

    # ---- device path: all-gather the shard partials where they lie (RCCL, in-stream) ----
    class _Alias:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False),
                                             "version": 2}

    def stream_handle(self):
        """The stream the library must launch on so that its kernels and the
        collective are ordered without a host wait (torch's current stream)."""
        if self.device.type != "cuda":
            return None
        return self._torch.cuda.current_stream(self.device).cuda_stream

    def allgather_device(self, src_ptr, dst_ptr, n):
        """dst[world][n] <- all-gather(src[n]) on raw device pointers (fp64)."""
        if self.device.type != "cuda":
            raise RuntimeError("device all-gather needs the nccl backend")
        src = self._torch.as_tensor(self._Alias(src_ptr, n), device=self.device)
        dst = self._torch.as_tensor(self._Alias(dst_ptr, n * self.world_size), device=self.device)
        self._dist.all_gather_into_tensor(dst, src)

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
