// ESSTempering.calculate_phi (smcnuts/tempering/adaptive_tempering.py:18-63) without a host round trip per trial
// temperature.
//
// The reference runs scipy.optimize.bisect (scipy/optimize/Zeros/bisect.c: xtol 2e-12, rtol 4 eps, 100 iterations) on
//     f(phi) = ESS(phi * loglik + logprior - base) - alpha N,
// one reduction over N per trial.  Round 2 launched that reduction and waited for it on the host ~18 times per SMC
// iteration (and all-gathered it as often between shards).  Here the bisection's OWN iteration runs on the device:
//   * a pass evaluates f at the 15 trial points of the next FOUR bisection steps at once -- the midpoint bisect.c
//     would try next and, for either outcome of every sign test, the ones after it (a depth-4 binary tree; each point
//     is formed by the very operations of bisect.c: dm *= 0.5; xm = xa + dm);
//   * a single-wavefront kernel then walks the tree with bisect.c's tests (fm * fa >= 0 moves xa; fm == 0 or
//     |dm| < xtol + rtol |xm| ends) and leaves the new bracket in device memory for the next pass.
// Pass 0 is the reference's opening: f(1) >= 0 returns 1; else f(phi_old), f(1) must differ in sign.  Eleven passes
// (1 + ceil(39 / 4)) are enqueued back to back; the host reads ONE result.  Shards all-gather the 15 x 4 shard partials
// per pass in the stream (smcn_temper_bisect_pass / _decide) -- 11 small collectives instead of ~18 x 2.
#pragma once
#include "smcn_weights.hpp"

namespace smcn {

constexpr int kTbNodes = 15;        // trial points per pass: the depth-4 tree of bisection midpoints
constexpr int kTbLevels = 4;
constexpr int kTbPasses = 1 + (100 + kTbLevels - 1) / kTbLevels;   // bisect.c gives up after 100 steps
// device state (doubles): bracket and status of the bisection
enum { TB_XA = 0, TB_DM, TB_FA, TB_DONE, TB_RESULT, TB_ERROR, TB_STEPS, TB_STATE };

// trial point `node` (heap order, 1-based: children 2 t and 2 t + 1 = xa kept / xa moved) of the bracket (xa, dm)
__device__ __forceinline__ double tb_node_point(double xa, double dm, int node) {
    int depth = 31 - __builtin_clz((unsigned)node);       // levels below the root
    double xm = 0.0;
    for (int k = depth; k >= 0; --k) {
        dm *= 0.5;
        xm = xa + dm;
        if (k > 0 && ((node >> (k - 1)) & 1)) xa = xm;    // that level's test moved the left end
    }
    return xm;
}
__device__ __forceinline__ double tb_point(const double* st, int pass, int t, double phi_old) {   // t = 0 .. kTbNodes - 1
    if (pass == 0) return t == 0 ? 1.0 : phi_old;          // adaptive_tempering.py:58,62: f(1), then the bracket's ends
    return tb_node_point(st[TB_XA], st[TB_DM], t + 1);
}

// (max, count at the max, sum of e^(v - max) over the others, sum of e^(2 (v - max))) of two parts, in that order
struct LseQuad { double mx, cnt, s1, s2; };
__device__ __forceinline__ LseQuad lse_merge(const LseQuad& a, const LseQuad& b) {
    if (b.mx == -kInf) return a;
    if (a.mx == -kInf) return b;
    const bool ahi = a.mx >= b.mx;
    const LseQuad& hi = ahi ? a : b;
    const LseQuad& lo = ahi ? b : a;
    if (hi.mx == lo.mx) return {hi.mx, hi.cnt + lo.cnt, hi.s1 + lo.s1, hi.s2 + lo.s2};
    const double sh = finite_d(hi.mx) ? hi.mx : 0.0, sl = finite_d(lo.mx) ? lo.mx : 0.0;
    const double f = exp(sl - sh);
    return {hi.mx, hi.cnt, hi.s1 + (lo.s1 + lo.cnt) * f, hi.s2 + lo.s2 * f * f};
}

// One pass: every block's (max, cnt, s1, s2) of the tempered log-weights at the pass's trial points.
// part: [gridDim.x][kTbNodes][4]
__global__ void __launch_bounds__(256) temper_multi_partial_kernel(const double* __restrict__ lpri, const double* __restrict__ llik,
                                                                   int64_t N, double phi_old, const double* st, int pass,
                                                                   double* part) {
    __shared__ double sh[4][kTbNodes][4];
    const int nn = pass == 0 ? 2 : kTbNodes;
    if (pass > 0 && st[TB_DONE] != 0.0) return;            // (block-uniform) the root is already known
    double phi[kTbNodes];
#pragma unroll
    for (int t = 0; t < kTbNodes; ++t) phi[t] = t < nn ? tb_point(st, pass, t, phi_old) : 1.0;
    // pass A: the block's maxima
    double mx[kTbNodes];
#pragma unroll
    for (int t = 0; t < kTbNodes; ++t) mx[t] = -kInf;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const double p0 = combine_lp(lpri[i], llik[i], 0.0), p1 = combine_lp(lpri[i], llik[i], 1.0);
        const double base = combine_lp(lpri[i], llik[i], phi_old);
#pragma unroll
        for (int t = 0; t < kTbNodes; ++t) {
            const double v = phi[t] * (p1 - p0) + p0 - base;     // temper_logw_kernel's expression
            mx[t] = (v > mx[t] || (v != v && mx[t] == mx[t])) ? v : mx[t];   // (a NaN sticks, as fmax would not)
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < kTbNodes; ++t) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double w = __shfl_xor(mx[t], o, 64);
            mx[t] = (w > mx[t] || (w != w && mx[t] == mx[t])) ? w : mx[t];
        }
        if (lane == 0) sh[wv][t][0] = mx[t];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kTbNodes; ++t) {
        double m = sh[0][t][0];
        for (int w = 1; w < 4; ++w) { const double o = sh[w][t][0]; m = (o > m || (o != o && m == m)) ? o : m; }
        mx[t] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int t = 0; t < kTbNodes; ++t) sh[0][t][0] = mx[t];   // the BLOCK's maximum, for the partial written below
    }
    // pass B: counts and sums against the block's maxima (lse_partial_kernel's arithmetic)
    double cnt[kTbNodes], s1[kTbNodes], s2[kTbNodes];
#pragma unroll
    for (int t = 0; t < kTbNodes; ++t) { cnt[t] = 0.0; s1[t] = 0.0; s2[t] = 0.0; }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const double p0 = combine_lp(lpri[i], llik[i], 0.0), p1 = combine_lp(lpri[i], llik[i], 1.0);
        const double base = combine_lp(lpri[i], llik[i], phi_old);
#pragma unroll
        for (int t = 0; t < kTbNodes; ++t) {
            if (t < nn) {
                const double v = phi[t] * (p1 - p0) + p0 - base;
                if (v != -kInf) {
                    const double shift = finite_d(mx[t]) ? mx[t] : 0.0;
                    const double e = exp(v - shift);
                    if (v == mx[t]) cnt[t] += 1.0;
                    else s1[t] += e;
                    s2[t] = fma(e, e, s2[t]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < kTbNodes; ++t) {
        cnt[t] = wave_sum(cnt[t]); s1[t] = wave_sum(s1[t]); s2[t] = wave_sum(s2[t]);
        if (lane == 0) { sh[wv][t][1] = cnt[t]; sh[wv][t][2] = s1[t]; sh[wv][t][3] = s2[t]; }
    }
    __syncthreads();
    if (threadIdx.x < kTbNodes) {
        const int t = threadIdx.x;
        double* o = part + ((int64_t)blockIdx.x * kTbNodes + t) * 4;
        o[0] = sh[0][t][0];       // (the block's maximum, stored by thread 0 above)
        o[1] = ((sh[0][t][1] + sh[1][t][1]) + sh[2][t][1]) + sh[3][t][1];
        o[2] = ((sh[0][t][2] + sh[1][t][2]) + sh[2][t][2]) + sh[3][t][2];
        o[3] = ((sh[0][t][3] + sh[1][t][3]) + sh[2][t][3]) + sh[3][t][3];
    }
}

// this shard's partial per trial point: block t merges the partials of trial point t -- every lane a strided share of the
// blocks, then a fixed butterfly over the lanes (the same order every run)
__global__ void __launch_bounds__(64) temper_multi_local_kernel(const double* part, int nblocks, const double* st, int pass,
                                                                double* local /*[kTbNodes][4]*/) {
    const int t = blockIdx.x, lane = threadIdx.x;
    LseQuad q{-kInf, 0.0, 0.0, 0.0};
    if (!(pass > 0 && st[TB_DONE] != 0.0)) {
        for (int b = lane; b < nblocks; b += 64) {
            const double* p = part + ((int64_t)b * kTbNodes + t) * 4;
            q = lse_merge(q, {p[0], p[1], p[2], p[3]});
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const LseQuad w{__shfl_xor(q.mx, o, 64), __shfl_xor(q.cnt, o, 64), __shfl_xor(q.s1, o, 64), __shfl_xor(q.s2, o, 64)};
            q = (lane & o) ? lse_merge(w, q) : lse_merge(q, w);     // (lower lane's part first on both sides: same bits)
        }
    }
    if (lane == 0) { local[t * 4 + 0] = q.mx; local[t * 4 + 1] = q.cnt; local[t * 4 + 2] = q.s1; local[t * 4 + 3] = q.s2; }
}

// the shards' partials merged in rank order, f at the trial points, then bisect.c's steps along the tree
__global__ void __launch_bounds__(64) temper_multi_decide_kernel(const double* gath /*[world][kTbNodes][4]*/, int world,
                                                                 double target /* alpha N */, double phi_old, int pass,
                                                                 double* st) {
    __shared__ double f[kTbNodes];
    const int t = threadIdx.x;
    if (pass > 0 && st[TB_DONE] != 0.0) return;
    if (t < kTbNodes) {
        LseQuad q{-kInf, 0.0, 0.0, 0.0};
        bool nan = false;
        for (int r = 0; r < world; ++r) {
            const double* p = gath + ((int64_t)r * kTbNodes + t) * 4;
            nan = nan || (p[0] != p[0]);
            q = lse_merge(q, {p[0], p[1], p[2], p[3]});
        }
        // combine_lse_partials (parallel.py): scipy's logsumexp, then sum wn^2 = s2 e^(2 (shift - ll))
        double fv;
        if (nan) {
            fv = __builtin_nan("");
        } else if (q.mx == -kInf) {
            fv = __builtin_nan("");                       // every weight -inf: ESS undefined (the reference's nan)
        } else {
            const double shift = finite_d(q.mx) ? q.mx : 0.0;
            const double sm = q.s1 == 0.0 ? q.s1 : q.s1 / q.cnt;
            const double ll = log1p(sm) + log(q.cnt) + q.mx;
            const double sum_wn2 = q.s2 * exp(2.0 * (shift - ll));
            fv = 1.0 / sum_wn2 - target;
        }
        f[t] = fv;
    }
    __syncthreads();
    if (t != 0) return;
    const double xtol = 2e-12, rtol = 8.881784197001252e-16;
    if (pass == 0) {
        const double f1 = f[0], fo = f[1];
        st[TB_ERROR] = 0.0; st[TB_STEPS] = 0.0; st[TB_DONE] = 0.0;
        if (f1 >= 0.0) { st[TB_RESULT] = 1.0; st[TB_DONE] = 1.0; return; }          // adaptive_tempering.py:58-59
        // bisect.c: f(xa), f(xb); a root at an end; different signs required
        if (fo == 0.0) { st[TB_RESULT] = phi_old; st[TB_DONE] = 1.0; return; }
        if (f1 == 0.0) { st[TB_RESULT] = 1.0; st[TB_DONE] = 1.0; return; }
        if (signbit(fo) == signbit(f1)) { st[TB_ERROR] = 1.0; st[TB_DONE] = 1.0; st[TB_RESULT] = __builtin_nan(""); return; }
        st[TB_XA] = phi_old; st[TB_DM] = 1.0 - phi_old; st[TB_FA] = fo;
        return;
    }
    double xa = st[TB_XA], dm = st[TB_DM];
    const double fa = st[TB_FA];
    int steps = (int)st[TB_STEPS], node = 1;
    for (int level = 0; level < kTbLevels; ++level) {
        dm *= 0.5;
        const double xm = xa + dm;
        const double fm = f[node - 1];
        ++steps;
        node = 2 * node;
        if (fm * fa >= 0.0) { xa = xm; node += 1; }
        if (fm == 0.0 || fabs(dm) < xtol + rtol * fabs(xm)) { st[TB_RESULT] = xm; st[TB_DONE] = 1.0; break; }
        if (steps >= 100) { st[TB_ERROR] = 2.0; st[TB_DONE] = 1.0; st[TB_RESULT] = __builtin_nan(""); break; }   // "Failed to converge"
    }
    st[TB_XA] = xa; st[TB_DM] = dm; st[TB_STEPS] = (double)steps;
}

}  // namespace smcn
