// NUTS proposal, second generation of the kernel for models whose state is
// replicated on the G lanes of a group (arma, PRMwCD): same algorithm and same
// results as nuts_kernel (smcn_nuts.hpp; reference smcnuts/proposal/nuts.py:34-175),
// restructured around what the in-kernel profile showed -- two thirds of the
// cycles went into the divergent tree bookkeeping, not into the gradient:
//
//  * per-particle INPUT and OUTPUT records (16-byte chunks, one per lane) instead
//    of strided [D][N] accesses; the next particle's record is prefetched while the
//    current tree is built, so the queue never exposes HBM latency;
//  * the slice variable's Exp(1) is drawn by the prep kernel (no log1p here);
//  * uniforms come from a 32-entry LDS ring refilled 16 at a time by ONE Philox
//    call per lane at a convergent point (a draw is one broadcast ds_read_b64);
//  * the top-level accept / U-turn of a doubling is one more level of the same
//    merge loop; merge probabilities are compared through the sign of one FMA
//    (u*den - n'' < 0) instead of a division;
//  * U-turn dot products are computed once and sign-flipped by direction;
//  * LDS vectors are moved as 16-byte accesses; the start of a tree and the start
//    of a doubling share one code path.
#pragma once
#include "smcn_nuts.hpp"

namespace smcn {

// ---- record layouts (doubles) ---------------------------------------------------
// input : [x0(VP), r0(VP), e0, pad]                    2*VP + 2
// output: [x'(VP), r'(VP), lpri1, llik1, lpri0, llik0, stats0, stats1]   2*VP + 6
//         stats0 = nleap | depth << 32, stats1 = ndraws | flags << 32 (bit patterns)
__host__ __device__ constexpr int n2_vp(int DL) { return (DL + 1) & ~1; }
__host__ __device__ constexpr int n2_in_doubles(int DL) { return 2 * n2_vp(DL) + 2; }
__host__ __device__ constexpr int n2_out_doubles(int DL) { return 2 * n2_vp(DL) + 6; }
// L = tree-stack levels kept in LDS (Model::N2_LDS_LEVELS); deeper levels live in a global
// overflow area of n2_ovf_doubles per resident group.
__host__ __device__ constexpr int n2_slot_doubles(int DL, int L = 10) {
    const int VP = n2_vp(DL);
    int n = n2_out_doubles(DL) + 6 * VP + L * 2 * VP + L * (2 * VP + 4) + 32;
    n = (n + 1) & ~1;
    // slot stride = an odd multiple of 4 banks: the (up to 16) groups of a wave land on disjoint
    // 4-bank sets.  (L = 10, DL = 4 keeps the 274 the kernel was tuned with.)
    if (L == 10) { while ((2 * n) % 64 != 36) n += 2; }
    else { while (n % 4 != 2) n += 2; }
    return n;
}
__host__ __device__ constexpr int n2_ovf_doubles(int DL, int L) {
    return (10 - L) * (2 * n2_vp(DL) + 2 * n2_vp(DL) + 4);
}

struct Nuts2Args {
    int64_t N;
    int64_t particle_base;
    const double* mdata;
    const double* in;   // [N][n2_in_doubles]
    double* out;        // [N][n2_out_doubles]
    unsigned int* queue;
    double eps, phi, delta_max;
    int max_depth;
    uint64_t seed;
    uint32_t iter;
    int B;              // consecutive transitions per particle (records [B][N][..]); 1 = one SMC iteration
    const double* tape;
    const int64_t* tape_off;
    unsigned long long* prof;
    double* ovf;        // overflow tree-stack levels, one area per resident group (models with N2_LDS_LEVELS < 10)
    const double* logw0 = nullptr;   // nuts3 with B > 1 and the forward L-kernel: the log-weights before the block;
                                     // transitions b < B-1 then leave COMPACT records [x'(VP), logw_b, stats0]
    // nuts3 with fewer lanes than particles (smcn_nuts3.hpp, QUEUE): a particle's block of B transitions is handed on in SEGMENTS of seg_len
    // transitions (0: the whole block is one job); a lane that ends segment s leaves (x', running log-weight) in
    // handover[particle][s] for the lane of the same wavefront that takes the particle's next segment.
    int seg_len = 0;
    int seg_tail = 0;        // > 0: the block's last seg_tail transitions are a segment of their own
    int seg_align = 1;       // a lane starts a job's first tree in an iteration whose number is a multiple of this (a power of two)
    unsigned long long* handover = nullptr;  // [N][segments - 1][VH + 1] 16-byte pairs
    int wide = 3;                    // nuts3: bit 0 = lane groups evaluate a wavefront's last stragglers (smcn_set_wide_eval:
                                     // re-associated sums); bit 1 = idle lanes draw the stragglers' uniforms (always on: same bits)
};

// prep: momentum draw (samples.py:155) + slice exponential (nuts.py:69) + packing
// of the input records.  r_in != null: momenta supplied by the caller.
__global__ void __launch_bounds__(256) nuts2_prep_kernel(const double* x, const double* r_in, double* r_out, double* in,
                                                         int64_t N, int D, int VP, int64_t particle_base, uint64_t seed,
                                                         uint32_t iter, int B, const double* tape,
                                                         const int64_t* tape_off) {
    // One thread builds one record (2 VP + 2 doubles: stride 80 B at D = 4); the block's 256 records
    // are contiguous in memory, so they are staged in LDS and leave as coalesced 16-byte chunks.
    extern __shared__ double prep_stage[];
    using d2 = double __attribute__((ext_vector_type(2)));
    const int RS = 2 * VP + 2;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x, t = t0 + threadIdx.x;
    const bool live = t < N * B;
    if (live) {
        const int b = (int)(t / N);
        const int64_t p = t - (int64_t)b * N;
        double* rec = prep_stage + (int64_t)threadIdx.x * RS;
        for (int c = 0; c < VP; ++c) rec[c] = (c < D && b == 0) ? x[(int64_t)c * N + p] : 0.0;
        if (r_in) {   // caller-supplied momenta (single transition only)
            for (int c = 0; c < VP; ++c) rec[VP + c] = (c < D) ? r_in[(int64_t)c * N + p] : 0.0;
        } else {
            for (int m = 0; 2 * m < VP; ++m) {
                double z0 = 0.0, z1 = 0.0;
                if (2 * m < D) {
                    const u32x4 o = philox4x32_10({(uint32_t)m, (uint32_t)(particle_base + p), iter + (uint32_t)b,
                                                   kStreamMomentum}, (uint32_t)seed, (uint32_t)(seed >> 32));
                    const double u1 = u53(o.a, o.b), u2 = u53(o.c, o.d);
                    const double rad = sqrt(-2.0 * log1p(-u1));
                    double sn, cs;
                    sincospi(2.0 * u2, &sn, &cs);        // exact argument; cheaper than sincos(2 pi u2)
                    z0 = rad * cs;
                    z1 = (2 * m + 1 < D) ? rad * sn : 0.0;
                    if (b == B - 1) {   // the resident r is the last transition's momentum
                        r_out[(int64_t)(2 * m) * N + p] = z0;
                        if (2 * m + 1 < D) r_out[(int64_t)(2 * m + 1) * N + p] = z1;
                    }
                }
                rec[VP + 2 * m] = z0;
                rec[VP + 2 * m + 1] = z1;
            }
        }
        double e0;
        if (tape) {
            const int64_t o = tape_off[p];
            e0 = (tape_off[p + 1] > o) ? tape[o] : 0.5;
        } else {
            e0 = -log1p(-philox_uniform(seed, iter + (uint32_t)b, (uint32_t)(particle_base + p), kStreamNuts, 0u));
        }
        rec[2 * VP] = e0;
        rec[2 * VP + 1] = 0.0;
    }
    __syncthreads();
    const int64_t nrec = (N * B - t0) < (int64_t)blockDim.x ? (N * B - t0) : (int64_t)blockDim.x;   // records of this block
    const int64_t nch = nrec * (RS / 2);
    d2* dst = reinterpret_cast<d2*>(in + t0 * RS);
    const d2* src = reinterpret_cast<const d2*>(prep_stage);
    for (int64_t c = threadIdx.x; c < nch; c += blockDim.x) dst[c] = src[c];
}

// The same for the lane kernel (smcn_nuts3.hpp), whose records are PAIR-MAJOR ([transition][16-byte pair][N]: consecutive
// threads write consecutive pairs, no staging): the start point only for the block's first transition (the kernel
// continues from its own samples), momentum and slice exponential for every transition.
template <int D>
__global__ void __launch_bounds__(256) nuts3_prep_kernel(const double* x, const double* r_in, double* r_out, double* in,
                                                         int64_t N, int64_t particle_base, uint64_t seed, uint32_t iter,
                                                         int B, const double* tape, const int64_t* tape_off) {
    using d2 = double __attribute__((ext_vector_type(2)));
    constexpr int VP = n2_vp(D), VH = VP / 2, IP = n2_in_doubles(D) / 2;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * B) return;
    const int b = (int)(t / N);
    const int64_t p = t - (int64_t)b * N;
    d2* const rec = reinterpret_cast<d2*>(in) + (int64_t)b * IP * N + p;      // pair k at rec[k * N]
    if (b == 0) {
#pragma unroll
        for (int k = 0; k < VH; ++k) {
            d2 v;
            v.x = x[(int64_t)(2 * k) * N + p];
            v.y = (2 * k + 1 < D) ? x[(int64_t)(2 * k + 1 < D ? 2 * k + 1 : 0) * N + p] : 0.0;
            rec[(int64_t)k * N] = v;
        }
    }
#pragma unroll
    for (int m = 0; m < VH; ++m) {
        d2 z;
        if (r_in) {   // caller-supplied momenta (single transition only)
            z.x = r_in[(int64_t)(2 * m) * N + p];
            z.y = (2 * m + 1 < D) ? r_in[(int64_t)(2 * m + 1 < D ? 2 * m + 1 : 0) * N + p] : 0.0;
        } else {      // the draws of nuts2_prep_kernel: same keys, same arithmetic
            const u32x4 o = philox4x32_10({(uint32_t)m, (uint32_t)(particle_base + p), iter + (uint32_t)b, kStreamMomentum},
                                          (uint32_t)seed, (uint32_t)(seed >> 32));
            const double u1 = u53(o.a, o.b), u2 = u53(o.c, o.d);
            const double rad = sqrt(-2.0 * log1p(-u1));
            double sn, cs;
            sincospi(2.0 * u2, &sn, &cs);
            z.x = rad * cs;
            z.y = (2 * m + 1 < D) ? rad * sn : 0.0;
            if (b == B - 1) {   // the resident r is the last transition's momentum
                r_out[(int64_t)(2 * m) * N + p] = z.x;
                if (2 * m + 1 < D) r_out[(int64_t)(2 * m + 1 < D ? 2 * m + 1 : 0) * N + p] = z.y;
            }
        }
        rec[(int64_t)(VH + m) * N] = z;
    }
    d2 e;
    if (tape) {
        const int64_t o = tape_off[p];
        e.x = (tape_off[p + 1] > o) ? tape[o] : 0.5;
    } else {
        e.x = -log1p(-philox_uniform(seed, iter + (uint32_t)b, (uint32_t)(particle_base + p), kStreamNuts, 0u));
    }
    e.y = 0.0;
    rec[(int64_t)(2 * VH) * N] = e;
}

// post: unpack the output records of the LAST transition to the [D][N] / [N] arrays and
// re-weight with the forward L-kernel in the same pass (samples.py:183-196,
// forward_lkernel.py:35, nuts.py:189 with the N(0, I) momentum proposal).  With B > 1
// transitions per particle every intermediate generation g = 1..B is written to
// gen_x[g-1] ([D][N]) / gen_logw[g-1] ([N]) and the per-generation leapfrog and
// "moved" counts are accumulated into cnt[2*b], cnt[2*b+1] (integers: exact in fp64).
// DT > 0: the dimension at compile time (the lane kernel's pair-major records: loops unroll, index arithmetic folds)
template <int DT = 0>
__global__ void __launch_bounds__(256) nuts2_post_kernel(const double* out, const double* in, const double* x0,
                                                         const double* logw, double* x_new, double* r_new,
                                                         double* lpri0, double* llik0, double* lpri1, double* llik1,
                                                         int32_t* nleap, int32_t* depth, int32_t* ndraws,
                                                         int32_t* flags, double* logw_new, double* gen_x,
                                                         double* gen_logw, double* cnt, int64_t N, int D_rt, int VP_rt,
                                                         int B, int compact = 0, int soa = 0) {
    const int D = DT > 0 ? DT : D_rt, VP = DT > 0 ? n2_vp(DT > 0 ? DT : 1) : VP_rt;
    // record element d of particle p: particle-major (rs doubles per record, `base` at the record area's start, record
    // index ri) or, soa, pair-major [record][pair][N]
    auto el = [&](const double* base, int64_t ri, int rs, int64_t pp, int d) -> double {
        return soa ? base[((ri * (rs / 2) + (d >> 1)) * N + pp) * 2 + (d & 1)] : base[(ri * N + pp) * rs + d];
    };
    __shared__ double sh[32];   // 4 waves x 2 U counts
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < N;
    double lw = (live && logw) ? logw[p] : 0.0;
    const double cst = 0.5 * D * kLog2Pi;
    // One thread walks its particle's B transitions: only the weight is a chain, so the records of
    // U transitions are fetched and unpacked together (U independent load streams in flight; with
    // 65 536 threads on the chip the kernel is latency-, not bandwidth-bound) and then folded into
    // the weight in order -- the arithmetic, and its rounding, is that of one transition at a time.
    constexpr int U = 4;
    for (int b0 = 0; b0 < B; b0 += U) {
        double c1[U], c0[U], Lk[U], qk[U], leaps[U], moved[U], lwc[U];
        bool isc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + u;
            c1[u] = c0[u] = Lk[u] = qk[u] = leaps[u] = moved[u] = lwc[u] = 0.0;
            isc[u] = compact && b < B - 1;      // the kernel already folded this transition into the weight
            if (live && b < B && isc[u]) {
                // compact records are dense: [b][N][VP + 2] behind the [N] full records of the last transition
                const double* cbase = out + N * (2 * VP + 6);
                bool all = true;
                for (int c = 0; c < D; ++c) {
                    const double xv = el(cbase, b, VP + 2, p, c);
                    const double xp = (b == 0) ? x0[(int64_t)c * N + p] : el(cbase, b - 1, VP + 2, p, c);
                    all = all && (xv != xp);
                    if (gen_x) gen_x[((int64_t)b * D + c) * N + p] = xv;
                }
                lwc[u] = el(cbase, b, VP + 2, p, VP);
                const unsigned long long s0 = (unsigned long long)__double_as_longlong(el(cbase, b, VP + 2, p, VP + 1));
                leaps[u] = (double)(s0 & 0xffffffffu);
                moved[u] = all ? 1.0 : 0.0;
            } else if (live && b < B) {
                // compact mode: the one full record per particle is that of the last transition, at [p]
                const int OS = 2 * VP + 6, IS = 2 * VP + 2;
                const int64_t ro = compact ? 0 : b;         // the full record's index
                auto rec = [&](int d) { return el(out, ro, OS, p, d); };
                double k0 = 0.0, k1 = 0.0;
                bool all = true;
                for (int c = 0; c < D; ++c) {
                    const double xv = rec(c), rv = rec(VP + c), r0 = el(in, b, IS, p, VP + c);
                    const double xp = (b == 0) ? x0[(int64_t)c * N + p]
                                               : (compact ? el(out + N * OS, b - 1, VP + 2, p, c) : el(out, b - 1, OS, p, c));
                    all = all && (xv != xp);
                    k0 = fma(r0, r0, k0);
                    k1 = fma(rv, rv, k1);
                    if (b == B - 1) { x_new[(int64_t)c * N + p] = xv; r_new[(int64_t)c * N + p] = rv; }
                    if (gen_x) gen_x[((int64_t)b * D + c) * N + p] = xv;
                }
                const double a1 = rec(2 * VP), b1 = rec(2 * VP + 1), a0 = rec(2 * VP + 2), bb0 = rec(2 * VP + 3);
                const unsigned long long s0 = (unsigned long long)__double_as_longlong(rec(2 * VP + 4));
                const unsigned long long s1 = (unsigned long long)__double_as_longlong(rec(2 * VP + 5));
                if (b == B - 1) {
                    lpri1[p] = a1; llik1[p] = b1; lpri0[p] = a0; llik0[p] = bb0;
                    nleap[p] = (int32_t)(s0 & 0xffffffffu);
                    depth[p] = (int32_t)(s0 >> 32);
                    ndraws[p] = (int32_t)(s1 & 0xffffffffu);
                    flags[p] = (int32_t)(s1 >> 32);
                }
                qk[u] = -0.5 * k0 - cst;
                Lk[u] = -0.5 * k1 - cst;
                c1[u] = combine_lp(a1, b1, 1.0);
                c0[u] = combine_lp(a0, bb0, 1.0);
                leaps[u] = (double)(s0 & 0xffffffffu);
                moved[u] = all ? 1.0 : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + u;
            if (b < B && live) {
                lw = isc[u] ? lwc[u] : lw + c1[u] - c0[u] + Lk[u] - qk[u];
                if (gen_logw) gen_logw[(int64_t)b * N + p] = lw;
                if (b == B - 1 && logw_new) logw_new[p] = lw;
            }
        }
        if (cnt) {   // the 2 U counts of this chunk in ONE block reduction (integers: exact in fp64)
            const int w = threadIdx.x >> 6;
#pragma unroll
            for (int u = 0; u < U; ++u) { leaps[u] = wave_sum(leaps[u]); moved[u] = wave_sum(moved[u]); }
            __syncthreads();
            if ((threadIdx.x & 63u) == 0) {
#pragma unroll
                for (int u = 0; u < U; ++u) { sh[w * 2 * U + 2 * u] = leaps[u]; sh[w * 2 * U + 2 * u + 1] = moved[u]; }
            }
            __syncthreads();
            if (threadIdx.x < 2 * U && b0 + (int)(threadIdx.x >> 1) < B) {
                const double v = ((sh[threadIdx.x] + sh[2 * U + threadIdx.x]) + sh[4 * U + threadIdx.x]) + sh[6 * U + threadIdx.x];
                atomicAdd(cnt + 2 * (b0 + (int)(threadIdx.x >> 1)) + (threadIdx.x & 1u), v);
            }
        }
    }
}

}  // namespace smcn
