// Device-resident SMC iteration: every scalar the loop of
// SMCSampler.sample() (smcnuts/smc_sampler.py:109-149) needs -- log-likelihood,
// ESS, the resample decision (samples.py:120), the estimates -- is produced and
// consumed on the device, so K iterations are enqueued without a host round trip.
// Shards exchange ONE vector of 4 + 2*Dc partials per iteration.
#pragma once
#include "smcn_weights.hpp"

namespace smcn {

// per-iteration record in the device history
enum : int { H_LL = 0, H_ESS = 1, H_RESAMPLED = 2, H_LEAPS = 3, H_MOVED = 4, H_PHI = 5, H_MEAN = 6 };
__host__ __device__ constexpr int hist_stride(int Dc) { return 6 + 2 * Dc; }
// step scalars
enum : int { SS_LL = 0, SS_FLAG = 1, SS_LOGWVAL = 2, SS_ESS = 3, SS_SHIFT = 8 };

// e_i = exp(logw_i - shift) (0 for -inf) -> work; partials [cnt(max), s1(non-max), s2(all)]
__global__ void __launch_bounds__(kRedBlock) lse_e_partial_kernel(const double* a, int64_t N, const double* maxp,
                                                                  double* work, double* part) {
    __shared__ double sh[4];
    const double mx = maxp[0];
    const double shift = finite_d(mx) ? mx : 0.0;
    double cnt = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kRedBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kRedBlock) {
        const double v = a[i];
        double e = 0.0;
        if (v != -kInf) {
            e = exp(v - shift);
            if (v == mx) cnt += 1.0;
            else s1 += e;
            s2 = fma(e, e, s2);
        }
        work[i] = e;
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

// part[c][b] = sum e * c(x)_c ; part[Dc + c][b] = sum e * (c(x)_c - shift_c)^2
__global__ void __launch_bounds__(kRedBlock) moment2_partial_kernel(const double* x, const double* e, int64_t N, int D,
                                                                    int model_id, const double* shift, double* part) {
    __shared__ double sh[4];
    for (int c = 0; c < D; ++c) {
        double sa = 0.0, sb = 0.0;
        const double sc = shift[c];
        for (int64_t i = (int64_t)blockIdx.x * kRedBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kRedBlock) {
            const double v = constrain_coord(model_id, c, D, x[(int64_t)c * N + i]);
            const double w = e[i], d = v - sc;
            sa = fma(w, v, sa);
            sb = fma(w * d, d, sb);
        }
        sa = block_sum(sa, sh);
        sb = block_sum(sb, sh);
        if (threadIdx.x == 0) {
            part[(int64_t)c * gridDim.x + blockIdx.x] = sa;
            part[(int64_t)(D + c) * gridDim.x + blockIdx.x] = sb;
        }
    }
}

// ---- generation statistics in two launches ------------------------------------------
// part[gen][q][b], q = 0: block max; 1: count of elements at that max; 2: sum exp(a - max_b)
// over the others; 3: sum exp(2(a - max_b)); 4..4+Dc: sum e c(x)_c; then sum e (c(x)_c - shift_c)^2.
// Every block works relative to ITS OWN maximum (one pass, no grid-wide dependency); blocks
// are then combined exactly like shards (gen_reduce_blocks_kernel, combine_ranks_kernel).
// blockIdx.y = generation: logw + y*gsl, x + y*gsx.
__global__ void __launch_bounds__(kRedBlock) gen_partials_kernel(const double* logw0, const double* x0, int64_t N,
                                                                 int D, int model_id, const double* shift,
                                                                 double* part0, int64_t gsl, int64_t gsx,
                                                                 double* work /* [N] or null: e_i kept for large D */) {
    __shared__ double sh[4];
    const double* logw = logw0 + (int64_t)blockIdx.y * gsl;
    const double* x = x0 + (int64_t)blockIdx.y * gsx;
    const int nb = gridDim.x, NQ = 4 + 2 * D;
    double* part = part0 + (int64_t)blockIdx.y * NQ * nb;
    double m = -kInf, nanflag = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kRedBlock + threadIdx.x; i < N; i += (int64_t)nb * kRedBlock) {
        const double v = logw[i];
        if (v != v) nanflag = 1.0;
        m = fmax(m, v);
    }
    m = block_max(m, sh);
    nanflag = block_max(nanflag, sh);
    const double mx = nanflag != 0.0 ? __builtin_nan("") : m;
    const double sft = finite_d(mx) ? mx : 0.0;
    double cnt = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kRedBlock + threadIdx.x; i < N; i += (int64_t)nb * kRedBlock) {
        const double v = logw[i];
        if (v == -kInf) continue;
        const double e = exp(v - sft);
        if (v == mx) cnt += 1.0;
        else s1 += e;
        s2 = fma(e, e, s2);
        if (work) work[i] = e;
    }
    cnt = block_sum(cnt, sh);
    s1 = block_sum(s1, sh);
    s2 = block_sum(s2, sh);
    if (threadIdx.x == 0 && blockIdx.z == 0) {
        part[blockIdx.x] = mx;
        part[nb + blockIdx.x] = cnt;
        part[2 * nb + blockIdx.x] = s1;
        part[3 * nb + blockIdx.x] = s2;
    }
    // large D: the coordinates are spread over blockIdx.z (every z-block redoes the cheap weight pass above); at
    // D = 256 one block walked 256 coordinates with two block reductions each: 0.33 ms for 268 MB
    const int cper = (D + (int)gridDim.z - 1) / (int)gridDim.z;
    const int c_lo = (int)blockIdx.z * cper, c_hi = (c_lo + cper < D) ? c_lo + cper : D;
    // A thread meets the SAME particles for every coordinate (i = its first + k * stride): with at most four of them their
    // weights e_i stay in registers across the coordinate loop instead of being re-read per coordinate (D = 256: the e_i
    // were a second 268 MB stream beside x -- 156 -> 9x us for the launch).  Same values, same order of the sums.
    constexpr int EK = 4;
    const int64_t first = (int64_t)blockIdx.x * kRedBlock + threadIdx.x, stride = (int64_t)nb * kRedBlock;
    if (N <= stride * EK) {
        double ev[EK];
        bool on[EK];
#pragma unroll
        for (int k = 0; k < EK; ++k) {
            const int64_t i = first + k * stride;
            double v = -kInf;
            if (i < N) v = logw[i];
            on[k] = i < N && v != -kInf;
            ev[k] = on[k] ? exp(v - sft) : 0.0;
        }
        for (int c = c_lo; c < c_hi; ++c) {
            double sa = 0.0, sb = 0.0;
            const double sc = shift[c];
            const double* const xc = x + (int64_t)c * N;
#pragma unroll
            for (int k = 0; k < EK; ++k) {
                if (on[k]) {
                    const double xv = constrain_coord(model_id, c, D, xc[first + k * stride]);
                    const double d = xv - sc;
                    sa = fma(ev[k], xv, sa);
                    sb = fma(ev[k] * d, d, sb);
                }
            }
            sa = block_sum(sa, sh);
            sb = block_sum(sb, sh);
            if (threadIdx.x == 0) {
                part[(int64_t)(4 + c) * nb + blockIdx.x] = sa;
                part[(int64_t)(4 + D + c) * nb + blockIdx.x] = sb;
            }
        }
        return;
    }
    for (int c = c_lo; c < c_hi; ++c) {
        double sa = 0.0, sb = 0.0;
        const double sc = shift[c];
        for (int64_t i = (int64_t)blockIdx.x * kRedBlock + threadIdx.x; i < N; i += (int64_t)nb * kRedBlock) {
            const double v = logw[i];
            if (v == -kInf) continue;
            const double e = work ? work[i] : exp(v - sft);   // same thread wrote work[i] above
            const double xv = constrain_coord(model_id, c, D, x[(int64_t)c * N + i]);
            const double d = xv - sc;
            sa = fma(e, xv, sa);
            sb = fma(e * d, d, sb);
        }
        sa = block_sum(sa, sh);
        sb = block_sum(sb, sh);
        if (threadIdx.x == 0) {
            part[(int64_t)(4 + c) * nb + blockIdx.x] = sa;
            part[(int64_t)(4 + D + c) * nb + blockIdx.x] = sb;
        }
    }
}
// blocks -> this shard's partials [max, cnt, s1, s2, A.., B..] (same format as before);
// one block per generation.
__global__ void __launch_bounds__(kRedBlock) gen_reduce_blocks_kernel(const double* part0, int nb, int Dc,
                                                                      double* out0) {
    __shared__ double sh[4];
    const int NQ = 4 + 2 * Dc;
    const double* part = part0 + (int64_t)blockIdx.x * NQ * nb;
    double* out = out0 + (int64_t)blockIdx.x * NQ;
    double M = -kInf, nanflag = 0.0;
    for (int b = threadIdx.x; b < nb; b += kRedBlock) {
        const double v = part[b];
        if (v != v) nanflag = 1.0;
        M = fmax(M, v);
    }
    M = block_max(M, sh);
    nanflag = block_max(nanflag, sh);
    if (nanflag != 0.0) M = __builtin_nan("");
    const double sM = finite_d(M) ? M : 0.0;
    double cnt = 0.0, s1 = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < nb; b += kRedBlock) {
        const double mb = part[b];
        if (mb == -kInf || mb != mb) continue;
        const double scale = exp((finite_d(mb) ? mb : 0.0) - sM);
        const double cb = part[nb + b], s1b = part[2 * nb + b];
        if (mb == M) { cnt += cb; s1 += s1b * scale; }
        else s1 += (s1b + cb) * scale;
        s2 += part[3 * nb + b] * scale * scale;
    }
    cnt = block_sum(cnt, sh);
    s1 = block_sum(s1, sh);
    s2 = block_sum(s2, sh);
    if (threadIdx.x == 0 && blockIdx.y == 0) { out[0] = M; out[1] = cnt; out[2] = s1; out[3] = s2; }
    // the moment sums are spread over blockIdx.y (each block redoes the cheap scalar part above): at D = 256 one
    // block walking 516 quantities x 1024 partials took 0.55 ms
    for (int q = 4 + blockIdx.y; q < NQ; q += gridDim.y) {
        double a = 0.0;
        for (int b = threadIdx.x; b < nb; b += kRedBlock) {
            const double mb = part[b];
            if (mb == -kInf || mb != mb) continue;
            a += part[(int64_t)q * nb + b] * exp((finite_d(mb) ? mb : 0.0) - sM);
        }
        a = block_sum(a, sh);
        if (threadIdx.x == 0) out[q] = a;
    }
}

// Combine the shard partials [max, cnt, s1, s2, A_0.., B_0..] in rank order
// (samples.py:96-113 through scipy's logsumexp; estimate.py:79-95 with the
// shifted one-pass variance), decide on resampling (samples.py:120), record.
__device__ __forceinline__ void combine_ranks_body(const double* gathered, int world, int rank, int Dc,
                                                   double n_total, double log_n_local, const double* shift,
                                                   double phi, double* hist_k, double* ss, int rank_stride) {
    // every thread of the (one-wave) block redoes the scalar part; the Dc coordinates are strided over the threads
    const int tid = threadIdx.x, nth = blockDim.x;
    const int NQ = rank_stride > 0 ? rank_stride : 4 + 2 * Dc;   // doubles between two ranks' blocks
    double M = -kInf;
    bool nan = false;
    for (int g = 0; g < world; ++g) {
        const double mg = gathered[g * NQ];
        if (mg != mg) nan = true;
        M = fmax(M, mg);
    }
    const double shiftM = finite_d(M) ? M : 0.0;
    double m = 0.0, s = 0.0, s2 = 0.0, W = 0.0;
    for (int g = 0; g < world; ++g) {
        const double* p = gathered + g * NQ;
        if (p[0] == -kInf) continue;
        const double sg = finite_d(p[0]) ? p[0] : 0.0;
        const double scale = exp(sg - shiftM);
        if (p[0] == M) { m += p[1]; s += p[2] * scale; }
        else s += (p[2] + p[1]) * scale;
        s2 += p[3] * scale * scale;
        W += (p[2] + p[1]) * scale;
    }
    const double sm = (s == 0.0) ? s : s / m;
    double ll = log1p(sm) + log(m) + M;
    if (nan) ll = __builtin_nan("");
    const double ess = 1.0 / (s2 * exp(2.0 * (shiftM - ll)));
    for (int c = tid; c < Dc; c += nth) {
        double A = 0.0, B = 0.0;
        for (int g = 0; g < world; ++g) {
            const double* p = gathered + g * NQ;
            if (p[0] == -kInf) continue;
            const double scale = exp((finite_d(p[0]) ? p[0] : 0.0) - shiftM);
            A += p[4 + c] * scale;
            B += p[4 + Dc + c] * scale;
        }
        const double mean = A / W;
        const double dm = mean - shift[c];
        hist_k[H_MEAN + c] = mean;
        hist_k[H_MEAN + Dc + c] = B / W - dm * dm;
        ss[SS_SHIFT + c] = mean;   // next iteration's shift
    }
    // this shard's own log-sum-exp, for local resampling: logw <- log W_shard - log N_local
    const double* q = gathered + rank * NQ;
    const double sq = (q[2] == 0.0) ? q[2] : q[2] / q[1];
    const double ll_local = log1p(sq) + log(q[1]) + q[0];
    const bool res = ess < 0.5 * n_total;
    if (tid != 0) return;
    hist_k[H_LL] = ll;
    hist_k[H_ESS] = ess;
    hist_k[H_RESAMPLED] = res ? 1.0 : 0.0;
    hist_k[H_PHI] = phi;
    ss[SS_LL] = ll;
    ss[SS_FLAG] = res ? 1.0 : 0.0;
    ss[SS_LOGWVAL] = ll_local - log_n_local;
    ss[SS_ESS] = ess;
}

__global__ void combine_ranks_kernel(const double* gathered, int world, int rank, int Dc, double n_total,
                                     double log_n_local, const double* shift, double phi, double* hist_k,
                                     double* ss, int rank_stride = 0) {
    if (blockIdx.x != 0) return;
    combine_ranks_body(gathered, world, rank, Dc, n_total, log_n_local, shift, phi, hist_k, ss, rank_stride);
}
// B generations of a fused block in ONE launch: block g combines generation g of the gathered
// partials ([world][B][nq]) into history row g.  The step scalars `ss` are left as the LAST
// generation's (what B successive launches of combine_ranks_kernel leave); the other blocks write
// theirs to a scratch row.
__global__ void combine_ranks_gens_kernel(const double* gathered, int world, int rank, int Dc, double n_total,
                                          double log_n_local, const double* shift, double phi, double* hist0,
                                          int hist_stride, double* ss, double* ss_scratch, int ss_stride) {
    const int g = blockIdx.x, B = gridDim.x, NQ = 4 + 2 * Dc;
    combine_ranks_body(gathered + (int64_t)g * NQ, world, rank, Dc, n_total, log_n_local, shift, phi,
                       hist0 + (int64_t)g * hist_stride, g == B - 1 ? ss : ss_scratch + (int64_t)g * ss_stride,
                       B * NQ);
}

__global__ void wn_dev_kernel(const double* logw, double* wn, int64_t N, const double* ss) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double v = logw[i];
    wn[i] = (v == -kInf) ? 0.0 : exp(v - ss[SS_LL]);
}

// conditional (device flag) variants of the resampling kernels
__global__ void __launch_bounds__(256) scan_tile_if_kernel(const double* ss, const double* w, int64_t N, double* local,
                                                           double* ttot) {
    if (ss[SS_FLAG] == 0.0) return;
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
__global__ void scan_offsets_if_kernel(const double* ss, const double* ttot, int nt, double* toff) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || ss[SS_FLAG] == 0.0) return;
    scan_offsets_body(ttot, nt, toff);
}
__global__ void __launch_bounds__(256) search_gather_if_kernel(const double* ss, const double* local, const double* toff,
                                                               int64_t N, const double* u, uint64_t seed, uint32_t iter,
                                                               int64_t particle_base, const double* x, double* x_out,
                                                               int D, double* logw, int scheme, int64_t* idx_out = nullptr,
                                                               int gather = 1, const double* ttot = nullptr, int nt = 0) {
    __shared__ double sh_toff[kFusedOffsetsMaxTiles + 1];
    if (ss[SS_FLAG] == 0.0) return;                  // (block-uniform)
    if (ttot) toff = tile_offsets_lds(ttot, nt, sh_toff);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double total = toff[(N - 1) / kScanTile] + local[N - 1];
    const double key = resample_key(scheme, u, i, particle_base + i, N, particle_base, seed, iter);
    const int64_t lo = cdf_search(key, total, toff, local, nt, N, ttot != nullptr);
    const int64_t src = lo < N ? lo : N - 1;
    if (gather)
        for (int c = 0; c < D; ++c) x_out[(int64_t)c * N + i] = x[(int64_t)c * N + src];
    if (idx_out) idx_out[i] = lo;
    logw[i] = ss[SS_LOGWVAL];
}
__global__ void copy_if_kernel(const double* ss, const double* src, double* dst, int64_t n) {
    if (ss[SS_FLAG] == 0.0) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
// out[0] = sum_b part[b]
__global__ void __launch_bounds__(kRedBlock) sum_to_kernel(const double* part, int nb, double* out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += kRedBlock) s += part[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) out[0] = s;
}

}  // namespace smcn
