"""ctypes binding of libsmcnuts_hip.so (include/smcnuts_hip.h).

There is no CPU fallback: if the library is missing or no GPU is visible the
product raises."""
import ctypes as C
import os

import numpy as np

from . import build as _build

MODEL_GAUSS, MODEL_ARMA, MODEL_PRMWCD, MODEL_HOST = 0, 1, 2, 3
HOST_TARGET_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_double), C.c_int,
                             C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                             C.POINTER(C.c_double))
LKERNEL_FORWARD, LKERNEL_GAUSSIAN = 0, 1

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
_ctx = C.c_void_p

SIGNATURES = {
    "smcn_version": ([], C.c_int),
    "smcn_last_error": ([_ctx], C.c_char_p),
    "smcn_ctx_create": ([C.POINTER(_ctx), C.c_int, C.c_int64, C.c_int64, C.c_int, _dp, C.c_int64], C.c_int),
    "smcn_ctx_destroy": ([_ctx], None),
    "smcn_dim": ([_ctx], C.c_int),
    "smcn_constrained_dim": ([_ctx], C.c_int),
    "smcn_fused_transitions": ([_ctx], C.c_int),
    "smcn_set_stream": ([_ctx, C.c_void_p], C.c_int),
    "smcn_synchronize": ([_ctx], C.c_int),
    "smcn_set_seed": ([_ctx, C.c_uint64], C.c_int),
    "smcn_set_state": ([_ctx, _dp, _dp], C.c_int),
    "smcn_get_state": ([_ctx, _dp, _dp, _dp], C.c_int),
    "smcn_get_proposal": ([_ctx, _dp, _dp, _dp, _dp], C.c_int),
    "smcn_set_momentum": ([_ctx, _dp], C.c_int),
    "smcn_set_proposal": ([_ctx, _dp, _dp, _dp], C.c_int),
    "smcn_target_eval": ([_ctx, _dp, C.c_int64, C.c_double, _dp, _dp, _dp, _dp], C.c_int),
    "smcn_target_constrain": ([_ctx, _dp, C.c_int64, _dp], C.c_int),
    "smcn_init_particles_std_normal": ([_ctx, C.c_double], C.c_int),
    "smcn_init_weights": ([_ctx, C.c_double, _dp], C.c_int),
    "smcn_normalise_partials": ([_ctx, _dp], C.c_int),
    "smcn_normalise_apply": ([_ctx, C.c_double], C.c_int),
    "smcn_normalise": ([_ctx, _dp, _dp], C.c_int),
    "smcn_moment_sums": ([_ctx, _dp, _dp], C.c_int),
    "smcn_resample_multinomial": ([_ctx, _dp, C.c_double, C.c_double, C.c_int64, _lp], C.c_int),
    "smcn_propose_nuts": ([_ctx, C.c_double, C.c_double, C.c_int, C.c_double, C.c_int64, _dp, _lp], C.c_int),
    "smcn_get_tree_stats": ([_ctx, _ip, _ip, _ip, _ip], C.c_int),
    "smcn_last_leapfrogs": ([_ctx, _lp], C.c_int),
    "smcn_get_density_parts": ([_ctx, _dp, _dp, _dp, _dp], C.c_int),
    "smcn_set_lkernel_values": ([_ctx, _dp, _dp], C.c_int),
    "smcn_reweight": ([_ctx, C.c_int], C.c_int),
    "smcn_accept_reject": ([_ctx, C.c_double, _dp, C.c_int64], C.c_int),
    "smcn_reweight_asymptotic": ([_ctx, C.c_double, C.c_double], C.c_int),
    "smcn_set_logw_density_ratio": ([_ctx, C.c_double, C.c_double], C.c_int),
    "smcn_gauss_lkernel_sums": ([_ctx, _dp, _dp], C.c_int),
    "smcn_gauss_lkernel_logpdf": ([_ctx, _dp, _dp, _dp, _dp, C.c_double], C.c_int),
    "smcn_gauss_lkernel_device": ([_ctx, _dp], C.c_int),
    "smcn_gauss_lkernel_buffers": ([_ctx, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)], C.c_int),
    "smcn_gauss_lkernel_stage": ([_ctx, C.c_int, C.c_int, C.c_double, _dp], C.c_int),
    "smcn_set_nuts_cap": ([_ctx, C.c_int, C.c_int], C.c_int),
    "smcn_set_nuts_requeue": ([_ctx, C.c_int], C.c_int),
    "smcn_nuts_parked": ([_ctx, C.POINTER(C.c_int64)], C.c_int),
    "smcn_temper_partials": ([_ctx, C.c_double, C.c_double, _dp], C.c_int),
    "smcn_temper_bisect": ([_ctx, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int)], C.c_int),
    "smcn_temper_bisect_pass": ([_ctx, C.c_int, C.c_double], C.c_int),
    "smcn_temper_bisect_buffers": ([_ctx, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)], C.c_int),
    "smcn_temper_bisect_decide": ([_ctx, C.c_int, C.c_int, C.c_double, C.c_double], C.c_int),
    "smcn_temper_bisect_result": ([_ctx, C.POINTER(C.c_double), C.POINTER(C.c_int)], C.c_int),
    "smcn_eval_proposed_parts": ([_ctx, C.c_int], C.c_int),
    "smcn_commit": ([_ctx, _lp], C.c_int),
    "smcn_fast_begin": ([_ctx, C.c_int64, C.c_int, C.c_int], C.c_int),
    "smcn_fast_buffers": ([_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int)], C.c_int),
    "smcn_set_resample_uniforms": ([_ctx, _dp], C.c_int),
    "smcn_step_begin": ([_ctx, C.c_int64], C.c_int),
    "smcn_step_finish": ([_ctx, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int,
                          C.c_double, C.c_int, C.c_int, _dp, _lp], C.c_int),
    "smcn_fast_read": ([_ctx, _dp, _dp, _dp], C.c_int),
    "smcn_fast_read_from": ([_ctx, _dp, _dp, _dp, C.c_int64], C.c_int),
    "smcn_history_download": ([_ctx, C.c_int64, C.c_int64, _dp, _dp], C.c_int),
    "smcn_fuse_begin": ([_ctx, C.c_int, C.c_int], C.c_int),
    "smcn_fuse_buffers": ([_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int)], C.c_int),
    "smcn_fuse_run": ([_ctx, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int,
                       C.c_double, C.c_int], C.c_int),
    "smcn_set_resample_scheme": ([_ctx, C.c_int], C.c_int),
    "smcn_set_wide_eval": ([_ctx, C.c_int], C.c_int),
    "smcn_set_lane_grid": ([_ctx, C.c_int64], C.c_int),
    "smcn_set_lane_segments": ([_ctx, C.c_int], C.c_int),
    "smcn_set_host_target": ([_ctx, HOST_TARGET_FN, C.c_void_p], C.c_int),
    "smcn_moment_sums_of": ([_ctx, _dp, C.c_int, _dp, _dp], C.c_int),
    "smcn_block_resample_local": ([_ctx, C.c_int64], C.c_int),
    "smcn_block_launch": ([_ctx, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double], C.c_int),
    "smcn_block_post": ([_ctx, C.c_int64, C.c_int, C.c_int], C.c_int),
    "smcn_block_partials_get": ([_ctx, C.c_int, _dp], C.c_int),
    "smcn_block_partials_set": ([_ctx, C.c_int, C.c_int, _dp], C.c_int),
    "smcn_block_stats": ([_ctx, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int], C.c_int),
    "smcn_block_wait": ([_ctx, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
    "smcn_block_commit": ([_ctx, C.c_int64, C.c_int], C.c_int),
    "smcn_block_ess": ([_ctx, C.c_int, _dp], C.c_int),
    "smcn_fuse_decide": ([_ctx, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_int)], C.c_int),
    "smcn_fuse_finish": ([_ctx, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                          C.POINTER(C.c_int)], C.c_int),
    "smcn_partials_get": ([_ctx, _dp], C.c_int),
    "smcn_partials_set_gathered": ([_ctx, _dp, C.c_int], C.c_int),
    "smcn_timers": ([_ctx, _dp, C.c_int], C.c_int),
    "smcn_selftest_math": ([_ctx, _dp, C.c_int64, _dp], C.c_int),
    "smcn_measure_peaks": ([_ctx, _dp], C.c_int),
    "smcn_selftest_wide": ([_ctx, C.c_int, _dp, C.c_int64, _dp], C.c_int),
    "smcn_device_cache_trim": ([C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)], C.c_int),
    "smcn_debug_profile": ([_ctx, C.POINTER(C.c_uint64), C.c_int], C.c_int),
    "smcn_bench_resample": ([_ctx, C.c_int, C.c_int64, _dp], C.c_int),
    "smcn_comm_unique_id": ([C.c_char_p], C.c_int),
    "smcn_comm_init": ([_ctx, C.c_int, C.c_int, C.c_char_p], C.c_int),
    "smcn_comm_destroy": ([_ctx], C.c_int),
    "smcn_comm_info": ([_ctx, C.POINTER(C.c_int)], C.c_int),
    "smcn_comm_allgather": ([_ctx, C.c_void_p, C.c_void_p, C.c_int64], C.c_int),
    "smcn_comm_allgather_host": ([_ctx, _dp, C.c_int64, _dp], C.c_int),
    "smcn_comm_alltoallv": ([_ctx, C.c_void_p, _lp, C.c_void_p, _lp, C.c_int], C.c_int),
    "smcn_buf_get": ([_ctx, C.c_void_p, C.c_int64, _dp], C.c_int),
    "smcn_buf_set": ([_ctx, C.c_void_p, C.c_int64, _dp], C.c_int),
    "smcn_buf_copy": ([_ctx, C.c_void_p, C.c_void_p, C.c_int64], C.c_int),
    "smcn_gres_begin": ([_ctx, C.c_int, _dp], C.c_int),
    "smcn_gres_buffers": ([_ctx] + [C.POINTER(C.c_void_p)] * 6, C.c_int),
    "smcn_gres_plan": ([_ctx, C.c_int, C.c_int, _dp, C.c_int64, _ip], C.c_int),
    "smcn_gres_set_order": ([_ctx, _ip], C.c_int),
    "smcn_gres_reserve": ([_ctx, C.c_int64], C.c_int),
    "smcn_gres_serve": ([_ctx, C.c_int, C.c_int, C.c_int64], C.c_int),
    "smcn_gres_finish": ([_ctx, C.c_int, _dp], C.c_int),
}

_lib = None


class SmcnError(RuntimeError):
    pass


def lib():
    """Load libsmcnuts_hip.so; raises if it has not been built."""
    global _lib
    if _lib is None:
        path = os.environ.get("SMCN_LIB") or _build.LIB     # SMCN_LIB: diagnostic / A-B builds of the same ABI
        if not os.path.exists(path):
            raise SmcnError(
                f"{path} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
        _lib = C.CDLL(path)
        for name, (args, res) in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.argtypes, fn.restype = args, res
    return _lib


def dptr(a):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


def iptr(a):
    return None if a is None else a.ctypes.data_as(_ip)


def lptr(a):
    return None if a is None else a.ctypes.data_as(_lp)


def trim_device_cache(device=-1):
    """Streams and device buffers of closed contexts stay pooled in the library (include/smcnuts_hip.h,
    smcn_device_cache_trim): give the idle buffers of `device` (-1: all) back to the driver.  Returns
    (bytes released, bytes that were idle)."""
    rel, idle = C.c_int64(0), C.c_int64(0)
    rc = lib().smcn_device_cache_trim(int(device), C.byref(rel), C.byref(idle))
    if rc:
        raise RuntimeError("smcn_device_cache_trim failed")
    return rel.value, idle.value


class Context:
    """One GPU shard of N particles (smcn_ctx)."""

    def __init__(self, n_particles, model_id, model_data, device=0, particle_base=0):
        self._lib = lib()
        self.N = int(n_particles)
        self.particle_base = int(particle_base)
        md = np.ascontiguousarray(model_data, dtype=np.float64)
        h = _ctx()
        rc = self._lib.smcn_ctx_create(C.byref(h), int(device), self.N, self.particle_base, int(model_id),
                                       dptr(md), md.size)
        if rc != 0:
            raise SmcnError(self._lib.smcn_last_error(None).decode())
        self._h = h
        self.D = self._lib.smcn_dim(h)
        self.Dc = self._lib.smcn_constrained_dim(h)
        self.fused_transitions = self._lib.smcn_fused_transitions(h) == 1

    def close(self):
        if getattr(self, "_h", None):
            self._lib.smcn_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call(self, name, *args):
        rc = getattr(self._lib, name)(self._h, *args)
        if rc != 0:
            raise SmcnError(f"{name}: {self._lib.smcn_last_error(self._h).decode()}")

    # ---- convenience wrappers -------------------------------------------------
    def set_seed(self, seed):
        self.call("smcn_set_seed", C.c_uint64(int(seed) & (2 ** 64 - 1)))

    def set_state(self, x=None, logw=None):
        x = None if x is None else np.ascontiguousarray(x, dtype=np.float64)
        logw = None if logw is None else np.ascontiguousarray(logw, dtype=np.float64)
        self.call("smcn_set_state", dptr(x), dptr(logw))

    def get_state(self, x=True, logw=True, wn=False):
        X = np.empty((self.N, self.D)) if x else None
        lw = np.empty(self.N) if logw else None
        w = np.empty(self.N) if wn else None
        self.call("smcn_get_state", dptr(X), dptr(lw), dptr(w))
        return X, lw, w

    def get_proposal(self, r=True, x_new=True, r_new=True, logw_new=False):
        R = np.empty((self.N, self.D)) if r else None
        Xn = np.empty((self.N, self.D)) if x_new else None
        Rn = np.empty((self.N, self.D)) if r_new else None
        lw = np.empty(self.N) if logw_new else None
        self.call("smcn_get_proposal", dptr(R), dptr(Xn), dptr(Rn), dptr(lw))
        return R, Xn, Rn, lw

    def target_eval(self, x, phi=1.0, want_grad=False, want_parts=False):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        M = x.shape[0]
        lp = np.empty(M)
        g = np.empty((M, self.D)) if want_grad else None
        a = np.empty(M) if want_parts else None
        b = np.empty(M) if want_parts else None
        self.call("smcn_target_eval", dptr(x), M, float(phi), dptr(lp), dptr(g), dptr(a), dptr(b))
        return lp, g, a, b

    def constrain(self, x):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        out = np.empty((x.shape[0], self.Dc))
        self.call("smcn_target_constrain", dptr(x), x.shape[0], dptr(out))
        return out

    def normalise_partials(self):
        p = np.empty(4)
        self.call("smcn_normalise_partials", dptr(p))
        return p

    def temper_partials(self, phi_old, phi_new):
        p = np.empty(4)
        self.call("smcn_temper_partials", float(phi_old), float(phi_new), dptr(p))
        return p

    def moment_sums(self, mean=None):
        s = np.empty(self.Dc)
        m = None if mean is None else np.ascontiguousarray(mean, dtype=np.float64)
        self.call("smcn_moment_sums", dptr(m), dptr(s))
        return s

    def resample(self, loglik, log_n_total, iteration, u=None, want_idx=False):
        u = None if u is None else np.ascontiguousarray(u, dtype=np.float64)
        idx = np.empty(self.N, dtype=np.int64) if want_idx else None
        self.call("smcn_resample_multinomial", dptr(u), float(loglik), float(log_n_total), int(iteration), lptr(idx))
        return idx

    def propose_nuts(self, step_size, phi, iteration, max_depth=10, delta_max=100.0, tape=None, tape_off=None):
        if tape is not None:
            tape = np.ascontiguousarray(tape, dtype=np.float64)
            tape_off = np.ascontiguousarray(tape_off, dtype=np.int64)
            if tape_off.size != self.N + 1:
                raise ValueError("tape_off must have N+1 entries")
        self.call("smcn_propose_nuts", float(step_size), float(phi), int(max_depth), float(delta_max),
                  int(iteration), dptr(tape), lptr(tape_off))

    def tree_stats(self):
        a, b, c, d = (np.empty(self.N, dtype=np.int32) for _ in range(4))
        self.call("smcn_get_tree_stats", iptr(a), iptr(b), iptr(c), iptr(d))
        return dict(nleap=a, depth=b, ndraws=c, flags=d)

    def last_leapfrogs(self):
        v = C.c_int64(0)
        self.call("smcn_last_leapfrogs", C.byref(v))
        return v.value

    def density_parts(self):
        a, b, c, d = (np.empty(self.N) for _ in range(4))
        self.call("smcn_get_density_parts", dptr(a), dptr(b), dptr(c), dptr(d))
        return a, b, c, d

    def commit(self, count_moved=True):
        v = C.c_int64(0)
        self.call("smcn_commit", C.byref(v) if count_moved else None)
        return v.value

    # ---- device-resident loop ----------------------------------------------------
    def fast_begin(self, K, save_history, world=1):
        self.call("smcn_fast_begin", int(K), int(bool(save_history)), int(world))
        a, b, n = C.c_void_p(), C.c_void_p(), C.c_int()
        self.call("smcn_fast_buffers", C.byref(a), C.byref(b), C.byref(n))
        self.lp_ptr, self.gath_ptr, self.nq = a.value, b.value, n.value

    def step_begin(self, k):
        self.call("smcn_step_begin", int(k))

    def step_finish(self, k, world, rank, n_total, step_size, phi, max_depth=10, delta_max=100.0, last=False,
                    tape=None, tape_off=None):
        if tape is not None:
            tape = np.ascontiguousarray(tape, dtype=np.float64)
            tape_off = np.ascontiguousarray(tape_off, dtype=np.int64)
        self.call("smcn_step_finish", int(k), int(world), int(rank), float(n_total), float(step_size), float(phi),
                  int(max_depth), float(delta_max), LKERNEL_FORWARD, int(bool(last)), dptr(tape), lptr(tape_off))

    def fast_read(self, K, save_history, xs=None, lw=None, k_from=0):
        """Scalar history and, with save_history, x_saved / logw_saved -- into the caller's arrays when given (already
        touched memory: a fresh 100 MB array costs more in page faults than its bytes cost on the bus)."""
        hs = 6 + 2 * self.Dc
        hist = np.empty((K + 1, hs))
        if save_history:
            ok = lambda a, shape: a is not None and a.shape == shape and a.dtype == np.float64 and a.flags.c_contiguous
            xs = xs if ok(xs, (K + 1, self.N, self.D)) else np.empty((K + 1, self.N, self.D))
            lw = lw if ok(lw, (K + 1, self.N)) else np.empty((K + 1, self.N))
        else:
            xs = lw = None
        self.call("smcn_fast_read_from", dptr(hist), dptr(xs), dptr(lw), int(k_from))
        return hist, xs, lw

    def partials_get(self):
        p = np.empty(self.nq)
        self.call("smcn_partials_get", dptr(p))
        return p

    def partials_set_gathered(self, g):
        g = np.ascontiguousarray(g, dtype=np.float64)
        self.call("smcn_partials_set_gathered", dptr(g), int(g.shape[0]))

    def timers(self, reset=False):
        t = np.zeros(6)
        self.call("smcn_timers", dptr(t), int(bool(reset)))
        return t
