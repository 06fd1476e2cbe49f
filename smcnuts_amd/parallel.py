"""Particle sharding over GPUs: one process per GPU, the only traffic is an
all-gather of a few doubles per reduction; every rank then
combines the shard partials in rank order, so all ranks hold bit-identical
scalars."""
import os
import time

import numpy as np

def _process_start_time():
    """When THIS process was started (not when this module was imported: a rank may spend a minute importing a framework
    before it gets here, while rank 0 has long written its id)."""
    try:        # field 22 of /proc/self/stat: start time in clock ticks since boot
        fields = open("/proc/self/stat").read().rsplit(")", 1)[1].split()
        age = float(open("/proc/uptime").read().split()[0]) - int(fields[19]) / os.sysconf("SC_CLK_TCK")
        return time.time() - max(age, 0.0)
    except (OSError, ValueError, IndexError):
        return time.time() - 600.0


_PROC_T0 = _process_start_time()      # an id file older than this process belongs to an earlier launch


def exchange_through_host(comm, ctx, send_ptr, send_counts, recv_ptr, recv_counts, elem):
    """All-to-all of device buffers for communicators that only offer a host all-gather (tests, gloo): every
    rank publishes its whole send buffer (padded to the longest) and picks its segments out of the others'."""
    from . import _capi
    W, r = comm.world_size, comm.rank
    send_counts, recv_counts = np.asarray(send_counts, dtype=np.int64), np.asarray(recv_counts, dtype=np.int64)
    totals = comm.allgather(np.array([float(send_counts.sum())]))[:, 0].astype(np.int64)
    cap = int(max(totals.max(), 1))
    buf = np.zeros(cap * elem)
    if send_counts.sum():
        ctx.call("smcn_buf_get", send_ptr, int(send_counts.sum()) * elem, _capi.dptr(buf))
    every = comm.allgather(buf)                                           # [W][cap * elem]
    counts = comm.allgather(send_counts.astype(np.float64)).astype(np.int64)   # counts[src][dst]
    parts = []
    for src in range(W):
        off = int(counts[src, :r].sum())
        assert counts[src, r] == recv_counts[src]
        parts.append(every[src, off * elem:(off + int(counts[src, r])) * elem])
    got = np.ascontiguousarray(np.concatenate(parts)) if parts else np.zeros(0)
    if got.size:
        ctx.call("smcn_buf_set", recv_ptr, got.size, _capi.dptr(got))


class SingleProcess:
    rank, world_size = 0, 1

    def allgather(self, v):
        return np.asarray(v, dtype=np.float64)[None, :]


class InProcessComm:
    """W shards inside ONE process, one host thread per shard, all on the GPU(s) this process sees -- the rehearsal of the
    shard protocol that a one-GPU box allows, and a way to run a population larger than one kernel's sweet spot as
    several contexts.  Every rank's view (`view(rank)`) offers the whole communicator interface, DEVICE PATH INCLUDED:
    the all-gather and the all-to-all move device buffers with device-to-device copies in the ranks' own streams
    (smcn_buf_copy), so the protocol runs exactly as over RcclComm -- device-side ESS bisection across shards, batched
    partials of fused blocks, routed global resampling, staged Gaussian L-kernel -- with a thread barrier where RCCL has
    its rendezvous."""

    def __init__(self, world, timeout=None):
        import threading
        self.world_size = int(world)
        # a rank thread that raises would leave the others waiting forever: every wait has a limit, and a rank that fails
        # inside a collective breaks the barrier for all (SMCN_INPROC_TIMEOUT seconds, default 600)
        self.timeout = float(os.environ.get("SMCN_INPROC_TIMEOUT", "600")) if timeout is None else float(timeout)
        self._bar = threading.Barrier(self.world_size)
        self._slots = [None] * self.world_size

    def abort(self):
        """Release every rank that waits in a collective (they raise): call from a rank thread that cannot go on."""
        self._bar.abort()

    def view(self, rank):
        return _InProcessRank(self, int(rank))


class _InProcessRank:
    device_path = True

    def __init__(self, group, rank):
        self._g, self.rank, self.world_size = group, rank, group.world_size
        self.ctx = None
        self.device_calls = dict(allgather=0, exchange=0)     # how often the device path was taken (tests)

    def attach(self, ctx):
        self.ctx = ctx
        return self

    def _wait(self):
        import threading
        try:
            self._g._bar.wait(self._g.timeout)
        except threading.BrokenBarrierError:
            raise RuntimeError(f"InProcessComm: rank {self.rank}: another rank failed or did not arrive within "
                               f"{self._g.timeout:.0f} s (SMCN_INPROC_TIMEOUT)") from None

    def _guard(self, fn):
        """Run this rank's part of a collective; a failure here releases the ranks that wait for it."""
        try:
            return fn()
        except BaseException:
            self._g._bar.abort()
            raise

    def _publish(self, item):
        g = self._g
        g._slots[self.rank] = item
        self._wait()
        got = list(g._slots)
        self._wait()
        return got

    def allgather(self, v):
        return np.stack(self._publish(np.array(v, dtype=np.float64)))

    def barrier(self):
        self._wait()

    def info(self):
        return dict(backend="in-process (device-to-device copies)", world_seen=self._g._bar.parties, rank_seen=self.rank,
                    rccl_version=None)

    def allgather_device(self, src_ptr, dst_ptr, n):
        """dst[world][n] <- every rank's src[n] (device pointers, fp64)."""
        src_ptr, dst_ptr, n = getattr(src_ptr, "value", src_ptr), getattr(dst_ptr, "value", dst_ptr), int(n)
        self._guard(lambda: self.ctx.call("smcn_synchronize"))     # my src is complete
        srcs = self._publish(src_ptr)

        def copies():
            for r, p in enumerate(srcs):
                self.ctx.call("smcn_buf_copy", dst_ptr + 8 * n * r, p, n)
            self.ctx.call("smcn_synchronize")                # my reads of the peers' buffers are done ...
        self._guard(copies)
        self._wait()                                         # ... before any of them is written again
        self.device_calls["allgather"] += 1

    def exchange(self, ctx, send_ptr, send_counts, recv_ptr, recv_counts, elem):
        """All-to-all of device buffers with per-peer counts (items of `elem` doubles), segments in rank order."""
        send_ptr, recv_ptr = getattr(send_ptr, "value", send_ptr), getattr(recv_ptr, "value", recv_ptr)
        sc = np.asarray(send_counts, dtype=np.int64)
        rc = np.asarray(recv_counts, dtype=np.int64)
        self._guard(lambda: ctx.call("smcn_synchronize"))
        peers = self._publish((send_ptr, sc))

        def copies():
            off = 0
            for src, (p, counts) in enumerate(peers):
                cnt = int(counts[self.rank])
                if cnt != int(rc[src]):
                    raise RuntimeError(f"InProcessComm.exchange: rank {self.rank} expects {int(rc[src])} items from rank {src}, "
                                       f"which sends {cnt}")
                if cnt:
                    ctx.call("smcn_buf_copy", recv_ptr + 8 * elem * off, p + 8 * elem * int(counts[:self.rank].sum()), cnt * elem)
                off += cnt
            ctx.call("smcn_synchronize")
        self._guard(copies)
        self._wait()
        self.device_calls["exchange"] += 1


class RcclComm:
    """The in-library communicator (include/smcnuts_hip.h: smcn_comm_*): RCCL over xGMI, driven from Python with
    ctypes only -- no torch.  Rank, world size and the rendezvous address come from the launcher's environment
    (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT as torch.distributed.run sets them); rank 0 creates the RCCL id
    and hands it to the other ranks of the node through a file keyed by the launch."""
    device_path = True

    def __init__(self, rank=None, world_size=None, addr=None, port=None, tag=None):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else int(world_size)
        self.addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        self.port = int(port if port is not None else os.environ.get("MASTER_PORT", "29500"))
        self.ctx = None
        if tag is None:                     # communicators are created in the same order on every rank
            RcclComm._created += 1
            tag = RcclComm._created
        self.tag = tag

    _created = 0

    def _id_path(self):
        """The id file of THIS launch: keyed by address, port, launcher process, the launcher's run id and restart count
        (a worker group restarted under the same agent never reads the previous group's id) and the communicator's
        ordinal."""
        import tempfile
        nonce = f"{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}"
        nonce = "".join(ch if ch.isalnum() or ch in "-_" else "-" for ch in nonce)
        key = f"smcn_rccl_{self.addr}_{self.port}_{os.getppid()}_{nonce}_{self.tag}.id"
        d = os.environ.get("SMCN_RENDEZVOUS_DIR")
        if d is None:                     # a directory of the user's own (0700), not a predictable name in the shared /tmp
            d = os.path.join(tempfile.gettempdir(), f"smcn-{os.getuid()}")
            os.makedirs(d, mode=0o700, exist_ok=True)
            st = os.stat(d)               # (exist_ok: somebody else may have made it first)
            if st.st_uid != os.getuid() or (st.st_mode & 0o077):
                raise RuntimeError(f"RcclComm: rendezvous directory {d} is not a private directory of this user "
                                   f"(owner {st.st_uid}, mode {st.st_mode & 0o777:o}); set SMCN_RENDEZVOUS_DIR")
        return os.path.join(d, key)

    @staticmethod
    def _launch_nonce():
        """What only the ranks of THIS launch agree on: the launcher process and the time it was started (a recycled pid
        has another start time).  Stored behind the id, so a reader decides by CONTENT whether a file is its launch's --
        no clock window that a staggered start could miss."""
        ppid = os.getppid()
        try:
            with open(f"/proc/{ppid}/stat", "rb") as f:
                start = f.read().rsplit(b")", 1)[1].split()[19].decode()      # field 22: starttime, in clock ticks
        except (OSError, IndexError):
            start = "na"
        return f"{ppid}:{start}".encode()

    def _share_id(self):
        """Rank 0 creates the RCCL id; the other ranks of this launch (one node) read it from a file in the
        rendezvous directory (no extra TCP port to collide on).  The file is removed once every rank has initialised."""
        import ctypes as C
        from . import _capi
        path = self._id_path()
        if self.rank == 0:
            buf = C.create_string_buffer(128)
            if _capi.lib().smcn_comm_unique_id(buf) != 0:
                raise _capi.SmcnError("smcn_comm_unique_id: " + _capi.lib().smcn_last_error(None).decode())
            if self.world_size > 1:
                for stale in (path, path + ".tmp"):      # what an earlier, failed launch with the same key left behind
                    try:
                        os.unlink(stale)
                    except OSError:
                        pass
                fd = os.open(path + ".tmp", os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
                with os.fdopen(fd, "wb") as f:
                    f.write(buf.raw + self._launch_nonce())
                os.replace(path + ".tmp", path)          # atomic: a reader never sees a partial id
            return buf.raw
        deadline = time.time() + float(os.environ.get("SMCN_RENDEZVOUS_TIMEOUT", "120"))
        nonce = self._launch_nonce()
        while True:
            try:
                # never the file of an older launch: the id is followed by the launch nonce (launcher pid + its start time),
                # which rank 0 of THIS launch wrote -- a file an earlier launch left under the same key carries another
                blob = open(path, "rb").read()
                if len(blob) > 128 and blob[128:] == nonce:
                    return blob[:128]
            except OSError:
                pass
            if time.time() > deadline:
                raise RuntimeError(f"RcclComm: no RCCL id from rank 0 at {path}")
            time.sleep(0.05)

    def attach(self, ctx):
        """Create the RCCL communicator on this context's device and stream (collective over all ranks).  ncclCommInitRank
        and the first collective have no time limit of their own: a watchdog thread ends THIS process (exit code 86, the
        reason on stderr) if they have not returned within SMCN_COMM_TIMEOUT seconds (default 180) -- a stuck rendezvous
        becomes a non-zero exit the launcher sees, not a hang.  (The process exits; it is never re-executed.)"""
        import sys
        import threading
        ident = self._share_id()
        done = threading.Event()
        limit = float(os.environ.get("SMCN_COMM_TIMEOUT", "180"))

        def watchdog():
            if not done.wait(limit):
                sys.stderr.write(f"RcclComm: rank {self.rank} of {self.world_size}: ncclCommInitRank / first collective "
                                 f"did not return within {limit:.0f} s (SMCN_COMM_TIMEOUT) -- giving up\n")
                sys.stderr.flush()
                os._exit(86)

        th = threading.Thread(target=watchdog, daemon=True)
        th.start()
        try:
            ctx.call("smcn_comm_init", self.rank, self.world_size, ident)
            self.ctx = ctx
            self.barrier()                  # the first collective: every rank has initialised
        finally:
            done.set()
        if self.rank == 0 and self.world_size > 1:
            try:
                os.unlink(self._id_path())  # every rank has read it
            except OSError:
                pass
        return self

    def stream_handle(self):
        return None        # the collectives already run in the context's own stream

    def allgather(self, v):
        from . import _capi
        v = np.ascontiguousarray(v, dtype=np.float64)
        out = np.empty((self.world_size,) + v.shape)
        self.ctx.call("smcn_comm_allgather_host", _capi.dptr(v.reshape(-1)), v.size, _capi.dptr(out.reshape(-1)))
        return out

    def allgather_device(self, src_ptr, dst_ptr, n):
        self.ctx.call("smcn_comm_allgather", src_ptr, dst_ptr, int(n))

    def exchange(self, ctx, send_ptr, send_counts, recv_ptr, recv_counts, elem):
        from . import _capi
        sc = np.ascontiguousarray(send_counts, dtype=np.int64)
        rc = np.ascontiguousarray(recv_counts, dtype=np.int64)
        ctx.call("smcn_comm_alltoallv", send_ptr, _capi.lptr(sc), recv_ptr, _capi.lptr(rc), int(elem))

    def barrier(self):
        self.allgather(np.zeros(1))

    def info(self):
        """What the communicator itself reports (ncclCommCount / ncclCommUserRank / ncclGetVersion), not the environment."""
        import ctypes as C
        v = (C.c_int * 3)()
        self.ctx.call("smcn_comm_info", v)
        return dict(backend="rccl-in-library", world_seen=int(v[0]), rank_seen=int(v[1]), rccl_version=int(v[2]))

    def close(self):
        if self.ctx is not None:
            self.ctx.call("smcn_comm_destroy")
            self.ctx = None


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

    def _side(self, ptr, n):
        """One side of an exchange as a tensor aliasing `n` doubles at `ptr`; a side with nothing to move may have no
        buffer at all (a rank that serves no requests) and becomes an empty tensor: every rank still joins the collective."""
        ptr = getattr(ptr, "value", ptr)
        if n == 0 or not ptr:
            if n:
                raise ValueError("exchange: a non-empty side without a buffer")
            return self._torch.empty(0, dtype=self._torch.float64, device=self.device)
        return self._torch.as_tensor(self._Alias(ptr, n), device=self.device)

    def exchange(self, ctx, send_ptr, send_counts, recv_ptr, recv_counts, elem):
        """All-to-all of device buffers with per-peer counts (items of `elem` doubles)."""
        if not self.device_path:
            return exchange_through_host(self, ctx, send_ptr, send_counts, recv_ptr, recv_counts, elem)
        ns, nr = int(np.sum(send_counts)) * elem, int(np.sum(recv_counts)) * elem

        src, dst = self._side(send_ptr, ns), self._side(recv_ptr, nr)
        self._dist.all_to_all_single(dst, src, [int(c) * elem for c in recv_counts], [int(c) * elem for c in send_counts])

    def barrier(self):
        self._dist.barrier()

    def info(self):
        return dict(backend=f"torch.distributed/{self._dist.get_backend()}" + ("" if self.device_path else " (host exchange)"),
                    world_seen=int(self._dist.get_world_size()), rank_seen=int(self._dist.get_rank()), rccl_version=None)

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
