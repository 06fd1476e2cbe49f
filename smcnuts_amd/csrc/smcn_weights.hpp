// Population kernels around the NUTS proposal: momentum / initial draws,
// batched target evaluation, log-weight normalisation + ESS partials,
// multinomial resampling (blocked scan + on-the-fly search + gather), weighted
// moments, re-weighting, tempering partials, commit.
//
// Replaces Samples.{normalise_weights, calculate_ess, _resample,
// propose_samples (momentum draw), _non_asympototic_reweight, _tempering,
// update_samples} (smcnuts/samples/samples.py:91-222), Estimate._estimate
// (smcnuts/estimate/estimate.py:79-95) and ESSTempering._ess
// (smcnuts/tempering/adaptive_tempering.py:41-56).  All are HBM/latency-bound
// streaming passes over [D][N] / [N] fp64 arrays.
#pragma once
#include "smcn_models.hpp"

namespace smcn {

constexpr int kScanTile = 1024;  // 256 threads x 4 consecutive elements

// ---- Box-Muller normals, [D][N], Philox (seed, iter, particle, stream) ------
__global__ void normals_kernel(double* out, int64_t N, int D, int64_t particle_base, uint64_t seed,
                               uint32_t iter, uint32_t stream) {
    const int npair = (D + 1) / 2;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * npair) return;
    const int m = (int)(t / N);
    const int64_t p = t - (int64_t)m * N;
    const u32x4 o = philox4x32_10({(uint32_t)m, (uint32_t)(particle_base + p), iter, stream},
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
    const double u1 = u53(o.a, o.b), u2 = u53(o.c, o.d);
    const double rad = sqrt(-2.0 * log1p(-u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);   // sin / cos of 2 pi u2 with an exact argument
    out[(int64_t)(2 * m) * N + p] = rad * cs;
    if (2 * m + 1 < D) out[(int64_t)(2 * m + 1) * N + p] = rad * sn;
}

// The same draws (same Philox keys, same Box-Muller) laid out particle-major, out[p][c]: the momentum of the targets whose
// particle fills a wavefront (nuts_wave_kernel reads a particle's row as coalesced 512-byte pieces); a pair is one 16-byte store
__global__ void normals_pm_kernel(double* out, int64_t N, int D, int64_t particle_base, uint64_t seed,
                                  uint32_t iter, uint32_t stream) {
    const int npair = (D + 1) / 2;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * npair) return;
    const int64_t p = t / npair;
    const int m = (int)(t - p * npair);
    const u32x4 o = philox4x32_10({(uint32_t)m, (uint32_t)(particle_base + p), iter, stream},
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
    const double u1 = u53(o.a, o.b), u2 = u53(o.c, o.d);
    double z0, z1;
    box_muller_lean(u1, u2, z0, z1);
    out[p * D + 2 * m] = z0;
    if (2 * m + 1 < D) out[p * D + 2 * m + 1] = z1;
}

// ---- batched target evaluation: one group of lanes per row ------------------
// x element (row i, coordinate c) at x[i*rs + c*cs].  Outputs may be null.
// grad is written with the same strides (grs, gcs).
template <class Model>
__global__ void __launch_bounds__(256) eval_kernel(const double* mdata, const double* x, int64_t M,
                                                   int64_t rs, int64_t cs, double phi, double* logp,
                                                   double* grad, int64_t grs, int64_t gcs, double* lpri_o,
                                                   double* llik_o) {
    constexpr int G = Model::G, DL = Model::DL;
    constexpr bool DIST = Model::DIST;
    extern __shared__ double eval_lds[];
    const int lg = (int)(threadIdx.x & (G - 1));
    Model model;
    model.init(mdata, lg, eval_lds);
    const int D = model.dim();
    const int64_t ngroups = (int64_t)gridDim.x * (blockDim.x / G);
    const int64_t g0 = (int64_t)blockIdx.x * (blockDim.x / G) + threadIdx.x / G;
    // every lane of a wavefront runs the same number of trips (eval is cooperative)
    const int64_t trips = (M + ngroups - 1) / ngroups;
    for (int64_t tr = 0; tr < trips; ++tr) {
        const int64_t i = g0 + tr * ngroups;
        const bool live = i < M;
        double xv[DL];
#pragma unroll
        for (int k = 0; k < DL; ++k) {
            const int c = DIST ? lg + G * k : k;
            xv[k] = (live && c < D) ? x[i * rs + c * cs] : 0.0;
        }
        double lpri, llik, gp[DL], gl[DL];
        model.eval(xv, lpri, llik, gp, gl);
        if (!live) continue;
        const double lp0 = lpri + phi * llik;
        const bool bad = !finite_d(lp0);
        if (lg == 0) {
            if (logp) logp[i] = bad ? -kInf : lp0;
            if (lpri_o) lpri_o[i] = lpri;
            if (llik_o) llik_o[i] = llik;
        }
        if (grad && (DIST || lg == 0)) {
#pragma unroll
            for (int k = 0; k < DL; ++k) {
                const int c = DIST ? lg + G * k : k;
                if (c < D) grad[i * grs + c * gcs] = bad ? -kInf : fma(phi, gl[k], gp[k]);
            }
        }
    }
}

// constrain(): exp() on the last coordinate for arma (sigma) / PRMwCD (Gamma)
__device__ __forceinline__ double constrain_coord(int model_id, int c, int D, double v) {
    // arma (1) / PRMwCD (2): the last coordinate is a log scale; Gaussian (0) and host-evaluated (3: the
    // caller constrains) are the identity
    return ((model_id == 1 || model_id == 2) && c == D - 1) ? exp(v) : v;
}
__global__ void constrain_kernel(const double* x, double* out, int64_t M, int D, int model_id) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * D) return;
    const int c = (int)(t % D);
    out[t] = constrain_coord(model_id, c, D, x[t]);
}

// ---- transposes between host [N][D] and device [D][N] ------------------------
__global__ void transpose_kernel(const double* in, double* out, int64_t rows, int64_t cols) {
    // in [rows][cols] -> out [cols][rows]
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * cols) return;
    const int64_t c = t / rows, r = t - c * rows;  // consecutive threads -> consecutive out
    out[t] = in[r * cols + c];
}

// ---- normalise_weights partials (samples.py:96-105; scipy logsumexp) ---------
// pass 1: per-block max over logw != -inf (NaN poisons, as np.max does)
__global__ void __launch_bounds__(kRedBlock) max_partial_kernel(const double* a, int64_t N, double* part) {
    __shared__ double sh[4];
    double m = -kInf, nanflag = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kRedBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kRedBlock) {
        const double v = a[i];
        if (v != v) nanflag = 1.0;
        m = fmax(m, v);
    }
    m = block_max(m, sh);
    nanflag = block_max(nanflag, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = nanflag != 0.0 ? __builtin_nan("") : m;
}
__global__ void __launch_bounds__(kRedBlock) max_final_kernel(const double* part, int nb, double* out) {
    __shared__ double sh[4];
    double m = -kInf, nanflag = 0.0;
    for (int i = threadIdx.x; i < nb; i += kRedBlock) {
        const double v = part[i];
        if (v != v) nanflag = 1.0;
        m = fmax(m, v);
    }
    m = block_max(m, sh);
    nanflag = block_max(nanflag, sh);
    if (threadIdx.x == 0) out[0] = nanflag != 0.0 ? __builtin_nan("") : m;
}
// pass 2: [count(a == max), sum_{a != max, a != -inf} exp(a - shift), sum_{a != -inf} exp(2 (a - shift))]
// with shift = max if finite else 0 (scipy).
__global__ void __launch_bounds__(kRedBlock) lse_partial_kernel(const double* a, int64_t N, const double* maxp,
                                                                double* part /*[3][nb]*/) {
    __shared__ double sh[4];
    const double mx = maxp[0];
    const double shift = finite_d(mx) ? mx : 0.0;
    double cnt = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kRedBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kRedBlock) {
        const double v = a[i];
        if (v == -kInf) continue;
        const double e = exp(v - shift);
        if (v == mx) cnt += 1.0;
        else s1 += e;
        s2 = fma(e, e, s2);
    }
    cnt = block_sum(cnt, sh);
    s1 = block_sum(s1, sh);
    s2 = block_sum(s2, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = cnt;
        part[gridDim.x + blockIdx.x] = s1;
        part[2 * gridDim.x + blockIdx.x] = s2;
    }
}
// sums `nv` vectors of nb block partials each: out[v] = sum_b part[v*nb + b], fixed order; one block per vector
// (launched with min(nv, 1024) blocks: the 377 sums of the Gaussian L-kernel took 230 us in a single block)
__global__ void __launch_bounds__(kRedBlock) sum_final_kernel(const double* part, int nb, int nv, double* out) {
    __shared__ double sh[4];
    for (int v = blockIdx.x; v < nv; v += gridDim.x) {
        double s = 0.0;
        for (int i = threadIdx.x; i < nb; i += kRedBlock) s += part[v * nb + i];
        s = block_sum(s, sh);
        if (threadIdx.x == 0) out[v] = s;
    }
}
// wn = exp(logw - loglik), 0 where logw == -inf   (samples.py:101-102)
__global__ void wn_kernel(const double* logw, double* wn, int64_t N, double loglik) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double v = logw[i];
    wn[i] = (v == -kInf) ? 0.0 : exp(v - loglik);
}

// ---- ESSTempering._ess logw at a trial temperature (adaptive_tempering.py:44)
__global__ void temper_logw_kernel(const double* lpri, const double* llik, double* out, int64_t N,
                                   double phi_old, double phi_new) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double p0 = combine_lp(lpri[i], llik[i], 0.0);
    const double p1 = combine_lp(lpri[i], llik[i], 1.0);
    const double base = combine_lp(lpri[i], llik[i], phi_old);
    out[i] = phi_new * (p1 - p0) + p0 - base;
}

// ---- weighted moments in constrained space (estimate.py:79-95) ----------------
// part[c][b] = sum over the block's particles of wn * f, f = c(x) (mean == null)
// or (c(x) - mean[c])^2.
__global__ void __launch_bounds__(kRedBlock) moment_partial_kernel(const double* x, const double* wn, int64_t N,
                                                                   int D, int model_id, const double* mean,
                                                                   double* part) {
    __shared__ double sh[4];
    for (int c = 0; c < D; ++c) {
        double s = 0.0;
        const double mc = mean ? mean[c] : 0.0;
        for (int64_t i = (int64_t)blockIdx.x * kRedBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kRedBlock) {
            double v = constrain_coord(model_id, c, D, x[(int64_t)c * N + i]);
            if (mean) { v -= mc; v = v * v; }
            s = fma(wn[i], v, s);
        }
        s = block_sum(s, sh);
        if (threadIdx.x == 0) part[c * gridDim.x + blockIdx.x] = s;
    }
}

// ---- multinomial resampling (samples.py:138-146) --------------------------------
// Inclusive scan of wn in the fixed blocked order (mirrored bit for bit by
// oracle.blocked_cumsum): per thread 4 consecutive elements sequentially,
// Hillis-Steele over the 64 lanes, the 4 wave totals of a tile sequentially.
__global__ void __launch_bounds__(256) scan_tile_kernel(const double* w, int64_t N, double* local, double* ttot) {
    __shared__ double wtot[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * 4;
    double s[4];
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double v = (base + k < N) ? w[base + k] : 0.0;
        acc = (k == 0) ? v : acc + v;
        s[k] = acc;
    }
    double v = s[3];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_up(v, o, 64);
        if (lane >= o) v = v + t;
    }
    double excl = __shfl_up(v, 1, 64);
    if (lane == 0) excl = 0.0;
    if (lane == 63) wtot[wv] = v;
    __syncthreads();
    double woff = 0.0;
    for (int k = 1; k <= wv; ++k) woff = woff + wtot[k - 1];
    const double off = woff + excl;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + k < N) local[base + k] = off + s[k];
    if (threadIdx.x == 255) ttot[blockIdx.x] = off + s[3];
}
// exclusive offsets of the tiles, sequentially; toff[nt] = total
// (one thread, left to right -- the order the tests' blocked reference sums in; the totals are fetched eight at a time so
// that the chain of additions does not wait for a memory round trip per tile)
__device__ __forceinline__ void scan_offsets_body(const double* __restrict__ ttot, int nt, double* __restrict__ toff) {
    double acc = 0.0;
    toff[0] = 0.0;
    int b = 0;
    for (; b + 8 <= nt; b += 8) {
        double t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = ttot[b + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc = acc + t[k]; toff[b + k + 1] = acc; }
    }
    for (; b < nt; ++b) { acc = acc + ttot[b]; toff[b + 1] = acc; }
}
__global__ void scan_offsets_kernel(const double* ttot, int nt, double* toff) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    scan_offsets_body(ttot, nt, toff);
}
// idx_i = searchsorted(cdf / cdf[-1], u_i, 'right') with cdf[j] = toff[tile(j)] + local[j]
// evaluated on the fly; then gather every coordinate.  u == null: Philox stream 2.
// Up to this many tiles (N <= 262 144 per shard) the search kernel sums the tile totals itself -- every block the same
// sequential sums as scan_offsets_body, in LDS -- and a resampling is TWO launches; beyond, the offsets come from the
// scan_offsets kernel (one thread's sequential pass would be longer than the launch it saves).
constexpr int kFusedOffsetsMaxTiles = 256;
__device__ __forceinline__ const double* tile_offsets_lds(const double* __restrict__ ttot, int nt, double* sh) {
    for (int b = (int)threadIdx.x; b < nt; b += (int)blockDim.x) sh[b + 1] = ttot[b];
    __syncthreads();
    if (threadIdx.x == 0) {
        double acc = 0.0;
        sh[0] = 0.0;
        for (int b = 0; b < nt; ++b) { acc = acc + sh[b + 1]; sh[b + 1] = acc; }   // (the association of scan_offsets_body)
    }
    __syncthreads();
    return sh;
}
// searchsorted(cdf, key, 'right') on cdf[i] = (toff[i / 1024] + local[i]) / total.  tiles_first (offsets in LDS): first
// the TILE (its last cdf value is toff[t + 1] / total, the very sum the element-wise search sees at that position), then
// the element inside it with three pivots per step -- 5 dependent global round trips instead of 16 (the search is
// latency-bound: 8 bytes per probe).
__device__ __forceinline__ int64_t cdf_search(double key, double total, const double* toff, const double* __restrict__ local,
                                              int nt, int64_t N, bool tiles_first) {
    int64_t lo = 0, hi = N;
    if (tiles_first) {
        int tl = 0, th = nt;
        while (tl < th) {
            const int tm = tl + ((th - tl) >> 1);
            if (key < toff[tm + 1] / total) th = tm;
            else tl = tm + 1;
        }
        if (tl < nt) {
            const double off = toff[tl];
            lo = (int64_t)tl * kScanTile;
            hi = lo + kScanTile < N ? lo + kScanTile : N;
            while (hi - lo > 7) {
                const int64_t w = hi - lo, p1 = lo + (w >> 2), p2 = lo + (w >> 1), p3 = lo + w - (w >> 2) - 1;   // lo <= p1 < p2 < p3 < hi
                const double c1 = (off + local[p1]) / total, c2 = (off + local[p2]) / total, c3 = (off + local[p3]) / total;
                if (key < c1) hi = p1;
                else if (key < c2) { lo = p1 + 1; hi = p2; }
                else if (key < c3) { lo = p2 + 1; hi = p3; }
                else lo = p3 + 1;
            }
            while (lo < hi) {
                const int64_t mid = lo + ((hi - lo) >> 1);
                if (key < (off + local[mid]) / total) hi = mid;
                else lo = mid + 1;
            }
        } else {
            lo = N;
        }
    } else {
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            const double cv = (toff[mid / kScanTile] + local[mid]) / total;
            if (key < cv) hi = mid;
            else lo = mid + 1;
        }
    }
    return lo;
}
// ttot != null (nt <= kFusedOffsetsMaxTiles): tile offsets from the tile totals, in LDS; else `toff` (global)
__global__ void __launch_bounds__(256) search_gather_kernel(const double* local, const double* toff, int nt, int64_t N,
                                                            const double* u, uint64_t seed, uint32_t iter,
                                                            int64_t particle_base, const double* x, double* x_out, int D,
                                                            double* logw, double logw_value, int64_t* idx_out, int scheme,
                                                            int gather = 1, const double* ttot = nullptr) {
    __shared__ double sh_toff[kFusedOffsetsMaxTiles + 1];
    if (ttot) toff = tile_offsets_lds(ttot, nt, sh_toff);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    // cdf[-1] exactly as the reference normalises: the last REAL element
    const double total = toff[(N - 1) / kScanTile] + local[N - 1];
    const double key = resample_key(scheme, u, i, particle_base + i, N, particle_base, seed, iter);
    const int64_t lo = cdf_search(key, total, toff, local, nt, N, ttot != nullptr);
    const int64_t src = lo < N ? lo : N - 1;
    if (gather)
        for (int c = 0; c < D; ++c) x_out[(int64_t)c * N + i] = x[(int64_t)c * N + src];
    logw[i] = logw_value;
    if (idx_out) idx_out[i] = lo;
}
// Second stage of the resampling for wide particles (D >= kGatherRowsMinD): x_out[c][i] = x[c][idx_i], one ROW c of
// the [D][N] layout at a time and rows dealt to the XCDs (blocks go round-robin over the 8 XCDs, so block b runs on
// XCD b % 8): the N * 8 bytes of a row stay in that XCD's L2 while all its particles are gathered from it, and HBM
// sees every row once.  (A thread walking all D rows of its particle touches D different rows between two uses of
// any of them: at D = 256, N = 131 072 that form moved 8-16x the bytes.)
// Source layout [world][D][n_src_local] (world = 1: the shard's own [D][N]); idx holds GLOBAL ancestor slots,
// possibly == n_src_total for a key beyond the last cdf value (clamped as searchsorted's caller does).
constexpr int kGatherRowsMinD = 16;
constexpr int kGatherPerThread = 4;
__global__ void __launch_bounds__(256) gather_rows_kernel(const double* flag, const int64_t* idx, int64_t n_src_total,
                                                          int64_t n_src_local, const double* __restrict__ x,
                                                          double* __restrict__ x_out, int64_t n_out, int D, int nchunks) {
    if (flag && *flag == 0.0) return;
    const int xcd = blockIdx.x & 7;
    const int q = blockIdx.x >> 3;
    const int c = (q / nchunks) * 8 + xcd;
    if (c >= D) return;
    const int64_t i0 = (int64_t)(q % nchunks) * (256 * kGatherPerThread) + threadIdx.x;
    double v[kGatherPerThread];
#pragma unroll
    for (int k = 0; k < kGatherPerThread; ++k) {
        const int64_t i = i0 + 256 * k;
        v[k] = 0.0;
        if (i < n_out) {
            int64_t src = idx[i];
            src = src < n_src_total ? src : n_src_total - 1;
            const int64_t sr = src / n_src_local, sl = src - sr * n_src_local;
            v[k] = x[((int64_t)sr * D + c) * n_src_local + sl];
        }
    }
#pragma unroll
    for (int k = 0; k < kGatherPerThread; ++k) {
        const int64_t i = i0 + 256 * k;
        if (i < n_out) x_out[(int64_t)c * n_out + i] = v[k];
    }
}

// ---- re-weighting (samples.py:183-196) with N(0, I) momentum proposal -------------
// logw_new = logw + pi_1(x_new) - pi_1(x) + L - q,  q = N(r; 0, I),
// L = N(-r_new; 0, I) (forward_lkernel.py:35) or the supplied per-particle Lg.
__global__ void reweight_kernel(const double* logw, const double* lpri0, const double* llik0, const double* lpri1,
                                const double* llik1, const double* r, const double* r_new, const double* Lg,
                                const double* qv, double* logw_new, int64_t N, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double k0 = 0.0, k1 = 0.0;
    for (int c = 0; c < D; ++c) {
        const double a = r[(int64_t)c * N + i], b = r_new[(int64_t)c * N + i];
        k0 = fma(a, a, k0);
        k1 = fma(b, b, k1);
    }
    const double cst = 0.5 * D * kLog2Pi;
    const double q = qv ? qv[i] : (-0.5 * k0 - cst);
    const double L = Lg ? Lg[i] : (-0.5 * k1 - cst);
    const double p_x = combine_lp(lpri0[i], llik0[i], 1.0);
    const double p_xn = combine_lp(lpri1[i], llik1[i], 1.0);
    logw_new[i] = logw[i] + p_xn - p_x + L - q;
}

// the same update from |r|^2 and |r'|^2 (forward L-kernel, N(0, I) momentum proposal)
__global__ void reweight_kin_kernel(const double* logw, const double* lpri0, const double* llik0, const double* lpri1,
                                    const double* llik1, const double* kin0, const double* kin1, double* logw_new,
                                    int64_t N, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double cst = 0.5 * D * kLog2Pi;
    const double q = -0.5 * kin0[i] - cst;
    const double L = -0.5 * kin1[i] - cst;
    const double p_x = combine_lp(lpri0[i], llik0[i], 1.0);
    const double p_xn = combine_lp(lpri1[i], llik1[i], 1.0);
    logw_new[i] = logw[i] + p_xn - p_x + L - q;
}

// ---- Gaussian approximation of the optimal L-kernel (gaussian_lkernel.py:45-82) ----
// Sums over particles of X = [-r_new, x_new] - shift and of the upper triangle
// of X X^T (the N-scaled part of np.mean / np.cov).  part[q][b], q < E + E(E+1)/2.
constexpr int kGlkSumsBlock = 1024;     // 16 wavefronts share a tile's sums (4 before round 5: 89 us per pass at N = 65 536, D = 13)
__global__ void __launch_bounds__(kGlkSumsBlock) glk_sums_kernel(const double* r_new, const double* x_new, int64_t N, int D,
                                                                 const double* shift, int TP, double* part) {
    extern __shared__ double sh[];
    const int E = 2 * D, nq = E + E * (E + 1) / 2;
    double* Xs = sh;             // [E][TP]
    double* acc = sh + E * TP;   // [nq]
    unsigned short* ab = reinterpret_cast<unsigned short*>(acc + nq);   // [nq] (a, b) of sum q: b = 0xFF for the single sums
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nt = blockDim.x, nw = nt >> 6;
    for (int q = tid; q < nq; q += nt) {
        acc[q] = 0.0;
        int a2 = q, b2 = 0xFF;
        if (q >= E) {
            int rem = q - E;
            a2 = 0;
            while (rem >= E - a2) { rem -= E - a2; ++a2; }
            b2 = a2 + rem;
        }
        ab[q] = (unsigned short)(a2 | (b2 << 8));
    }
    const int64_t ntiles = (N + TP - 1) / TP;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
        for (int t = tid; t < D * TP; t += nt) {
            const int c = t / TP, k = t - c * TP;
            const int64_t i = tile * TP + k;
            Xs[c * TP + k] = (i < N) ? (-r_new[(int64_t)c * N + i] - shift[c]) : 0.0;
            Xs[(D + c) * TP + k] = (i < N) ? (x_new[(int64_t)c * N + i] - shift[D + c]) : 0.0;
        }
        __syncthreads();
        // every sum as before (a lane's terms in k order, the wavefront's butterfly, one add into the block's accumulator);
        // four of them in flight per wavefront so that the LDS reads and the butterflies of one hide behind the others'
        auto one = [&](int q) __attribute__((always_inline)) -> double {
            const int a2 = ab[q] & 0xFF, b2 = ab[q] >> 8;
            double s = 0.0;
            for (int k = lane; k < TP; k += 64)
                s += (b2 == 0xFF) ? Xs[a2 * TP + k] : Xs[a2 * TP + k] * Xs[b2 * TP + k];
            return s;
        };
        int q = wv;
        for (; q + 3 * nw < nq; q += 4 * nw) {
            double s0 = one(q), s1 = one(q + nw), s2 = one(q + 2 * nw), s3 = one(q + 3 * nw);
            s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
            if (lane == 0) { acc[q] += s0; acc[q + nw] += s1; acc[q + 2 * nw] += s2; acc[q + 3 * nw] += s3; }
        }
        for (; q < nq; q += nw) {
            const double s = wave_sum(one(q));
            if (lane == 0) acc[q] += s;
        }
    }
    __syncthreads();
    for (int q = tid; q < nq; q += nt) part[(int64_t)q * gridDim.x + blockIdx.x] = acc[q];
}
// L_i = c0 - 0.5 | U^T ( -r_i - m0 - B (x_i - mu_x) ) |^2; par = [mu_x(D), m0(D), B(D*D), U(D*D)]
// (c0 by value, or -- c0p set -- read from device memory: the device-side algebra leaves it behind par)
__global__ void __launch_bounds__(256) glk_logpdf_kernel(const double* r_new, const double* x_new, int64_t N, int D,
                                                         const double* par, double c0, double* L, const double* c0p = nullptr) {
    extern __shared__ double sh[];
    if (c0p) {
        if (c0p[1] != 0.0) return;       // the algebra refused (status behind c0): L keeps what the caller had set
        c0 = *c0p;
    }
    double* P = sh;                      // 2D + 2D^2 parameters
    double* V = sh + 2 * D + 2 * D * D;  // [D][256] residuals of this block's particles
    const int tid = threadIdx.x;
    for (int t = tid; t < 2 * D + 2 * D * D; t += 256) P[t] = par[t];
    __syncthreads();
    const double *mux = P, *m0 = P + D, *B = P + 2 * D, *U = P + 2 * D + D * D;
    const int64_t i = (int64_t)blockIdx.x * 256 + tid;
    if (i >= N) return;
    for (int c = 0; c < D; ++c) {
        double m = m0[c];
        for (int d = 0; d < D; ++d) m = fma(B[c * D + d], x_new[(int64_t)d * N + i] - mux[d], m);
        V[c * 256 + tid] = -r_new[(int64_t)c * N + i] - m;
    }
    double maha = 0.0;
    for (int e = 0; e < D; ++e) {
        double z = 0.0;
        for (int c = 0; c < D; ++c) z = fma(V[c * 256 + tid], U[c * D + e], z);
        maha = fma(z, z, maha);
    }
    L[i] = c0 - 0.5 * maha;
}

// ---- asymptotic strategy (smcnuts/proposal/nuts_acc_rej.py:42-49, proposal/utils.py:3-34) -----
// Metropolis accept/reject of the NUTS move: reject iff u > min(1, exp(H1 - H0)) or x' has an
// infinite coordinate; a rejected particle keeps (x, r) and its density parts.
__device__ __forceinline__ uint32_t stream_accept() { return 4u; }
__global__ void accept_reject_kernel(const double* x, const double* r, double* x_new, double* r_new,
                                     const double* lpri0, const double* llik0, double* lpri1, double* llik1,
                                     const double* u, uint64_t seed, uint32_t iter, int64_t particle_base, double phi,
                                     int64_t N, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double k0 = 0.0, k1 = 0.0;
    bool inf = false;
    for (int c = 0; c < D; ++c) {
        const double a = r[(int64_t)c * N + i], b = r_new[(int64_t)c * N + i], xv = x_new[(int64_t)c * N + i];
        k0 = fma(a, a, k0);
        k1 = fma(b, b, k1);
        inf = inf || __builtin_isinf(xv);
    }
    const double H1 = combine_lp(lpri1[i], llik1[i], phi) - 0.5 * k1;
    const double H0 = combine_lp(lpri0[i], llik0[i], phi) - 0.5 * k0;
    const double ratio = exp(H1 - H0);
    const double prob = (ratio < 1.0) ? ratio : 1.0;   // Python's min(1., ratio): NaN -> 1.0
    const double ui = u ? u[i] : philox_uniform(seed, iter, (uint32_t)(particle_base + i), stream_accept(), 0u);
    if ((ui > prob) || inf) {
        for (int c = 0; c < D; ++c) {
            x_new[(int64_t)c * N + i] = x[(int64_t)c * N + i];
            r_new[(int64_t)c * N + i] = r[(int64_t)c * N + i];
        }
        lpri1[i] = lpri0[i];
        llik1[i] = llik0[i];
    }
}
// samples.py:169-180: logw_new = logw + pi_{phi_new}(x) - pi_{phi_old}(x) at the OLD positions
__global__ void reweight_asymptotic_kernel(const double* logw, const double* lpri0, const double* llik0,
                                           double* logw_new, int64_t N, double phi_old, double phi_new) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    logw_new[i] = logw[i] + combine_lp(lpri0[i], llik0[i], phi_new) - combine_lp(lpri0[i], llik0[i], phi_old);
}
// estimate_from_tempered.py:47: logw = pi_{a}(x) - pi_{b}(x) from stored density parts
__global__ void density_ratio_kernel(const double* lpri, const double* llik, double* logw, int64_t N, double pa,
                                     double pb) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    logw[i] = combine_lp(lpri[i], llik[i], pa) - combine_lp(lpri[i], llik[i], pb);
}

// ---- acceptance statistic (smc_sampler.py:97): all coordinates changed ------------
__global__ void __launch_bounds__(kRedBlock) moved_partial_kernel(const double* x, const double* x_new, int64_t N,
                                                                  int D, double* part) {
    __shared__ double sh[4];
    double cnt = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kRedBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kRedBlock) {
        bool all = true;
        for (int c = 0; c < D; ++c) all = all && (x_new[(int64_t)c * N + i] != x[(int64_t)c * N + i]);
        cnt += all ? 1.0 : 0.0;
    }
    cnt = block_sum(cnt, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = cnt;
}
__global__ void __launch_bounds__(kRedBlock) isum_partial_kernel(const int32_t* v, int64_t N, double* part) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kRedBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kRedBlock)
        s += (double)v[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// device-math self test: out[0][i] = exp_fast(x), out[1][i] = log1p_pos(|x|), out[2][i] = rcp_nr(x)
//                         out[3][i] = sum over i's wavefront of x (group_sum<64>), out[4][i] = the same butterfly with
//                         its last two stages through ds_bpermute (must agree bit for bit)
__global__ void selftest_math_kernel(const double* x, int64_t n, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double v = i < n ? x[i] : 0.0;
    const double s_swap = group_sum<64>(v), s_perm = wave_sum_bpermute(v);
    double a2, b2, a4, b4, c4, d4;            // the fused butterflies: sums of v, v^2 (and |v|, 1 - v) over the wavefront
    wave_sum2(v, v * v, a2, b2);
    wave_sum4(v, v * v, fabs(v), 1.0 - v, a4, b4, c4, d4);
    if (i >= n) return;
    double inv;
    out[i] = exp_fast(v);
    out[n + i] = log1p_pos(fabs(v), inv);
    out[2 * n + i] = rcp_nr(v);
    out[3 * n + i] = s_swap;
    out[4 * n + i] = s_perm;
    out[5 * n + i] = a2; out[6 * n + i] = b2;
    out[7 * n + i] = a4; out[8 * n + i] = b4; out[9 * n + i] = c4; out[10 * n + i] = d4;
}

// logw = lp - logq0 (samples.py:85)
__global__ void init_logw_kernel(const double* lp, const double* logq0, const double* x, double* logw, int64_t N,
                                 int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double lq;
    if (logq0) {
        lq = logq0[i];
    } else {  // N(x; 0, I)
        double ss = 0.0;
        for (int c = 0; c < D; ++c) { const double v = x[(int64_t)c * N + i]; ss = fma(v, v, ss); }
        lq = -0.5 * ss - 0.5 * D * kLog2Pi;
    }
    logw[i] = lp[i] - lq;
}

}  // namespace smcn
