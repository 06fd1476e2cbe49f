// NUTS proposal, third generation: ONE LANE OWNS ONE PARTICLE.
//
// Same algorithm, same draws and same results as nuts_kernel / nuts2_kernel (reference
// smcnuts/proposal/nuts.py:34-175: rvs, generate_nuts_samples, build_tree, NUTSLeapfrog,
// stop_criterion), for models whose value + gradient a single lane can evaluate without
// cross-lane traffic (LaneModel concept below: arma by forward sensitivities).
//
// Why: the arma kernel is bound by fp64 VALU ISSUE, not by HBM.  With G lanes per
// particle the T-step recurrence needs a two-pass scan (2 x the FMAs) and the whole per-particle
// scalar work -- density tail, leapfrog, tree state machine -- is issued once per 64/G particles.
// With one particle per lane the recurrence is the plain serial loop (10 fp64 instructions per
// time step, y_t in SGPRs, no scan, no DPP) and every other instruction is issued once per 64
// particles.  64 trees advance in lock step through the evaluation; the divergent tree
// bookkeeping is predicated per lane.
//
// Occupancy: N = 65 536 particles are 1 024 wavefronts = ONE per SIMD of the chip.  A wavefront
// therefore owns a quarter of its CU's LDS (40 KB = 640 B per lane) and the whole register file:
//   * LDS, lane-private, laid out [16-byte pair][lane] (conflict-free b128 accesses at any mix of
//     levels): accepted sample (x', r', density parts), the edge that is NOT moving (x, r, grad;
//     the moving edge is the live register state, so one slot serves both edges), an 8-entry
//     ring of Philox uniforms, the 10 sub-tree counts n' (u16), and the tree-stack levels that
//     are touched most: candidates of levels 0..LC-1 and first leaves of levels 1..LF
//     (LC = 3, LF = 2 at D = 4: 7/8 of all parks and merges);
//   * deeper stack levels in a global overflow area, [pair][lane] per wavefront (coalesced).
// There is no work queue: lane l of block b owns particle 64 b + l for all B fused transitions.
#pragma once
#include "smcn_nuts2.hpp"

namespace smcn {

constexpr int kN3Block = 64;   // one wavefront per block: no barrier, no coupling between waves

// ---------------------------------------------------------------------------------------------
// LaneModel concept:
//   static constexpr int D
//   void init(const double* mdata)                 wave-uniform set-up (scalar registers)
//   void eval(x[D], lpri, llik, gpri[D], glik[D])  this lane's particle, no cross-lane traffic
// ---------------------------------------------------------------------------------------------

// ARMA(1,1) (stan_models/arma/arma.stan:8-30), any series length T >= 1.  x = (mu, beta, theta, s),
// sigma = exp(s); mdata = [T, y_1..y_T].  Forward sensitivities of err_t:
//   err_t = y_t - mu - beta y_{t-1} - theta err_{t-1}          (t = 1: y_0 := mu, err_0 := 0)
//   d err_t / d(mu, beta, theta) = -(1, y_{t-1}, err_{t-1}) - theta * d err_{t-1}
// ten fp64 instructions per time step; y_t is wave-uniform and comes from scalar loads.
struct ArmaLaneModel {
    static constexpr int D = 4;
    using cptr = const __attribute__((address_space(4))) double*;
    int T;
    cptr y;

    __device__ __forceinline__ void init(const double* md) {
        T = (int)((cptr)md)[0];
        y = (cptr)md + 1;
    }

    __device__ __forceinline__ void eval(const double (&x)[4], double& lpri, double& llik, double (&gp)[4],
                                         double (&gl)[4]) const {
        const double mu = x[0], beta = x[1], theta = x[2], s = x[3];
        const double nth = -theta, nbeta = -beta;
        // The series is read 8 steps at a time into scalar registers, one chunk AHEAD of its use (a
        // scalar load issued and consumed in the same chunk exposes its latency to the only wave of
        // the SIMD); the buffer is padded by 32 doubles, so the look-ahead never leaves it.
        // t = 1 (arma.stan:25: nu_1 = mu + beta * mu)
        double c0 = y[0];
        double err = fma(nbeta, mu, c0 - mu);
        double dm = -(1.0 + beta), db = -mu, dt = 0.0;
        double ss = err * err, gm = err * dm, gb = err * db, gt = 0.0;
        auto step = [&](double yp, double yt) {   // t >= 2: yp = y_{t-1}, yt = y_t
            dt = fma(nth, dt, -err);                 // uses err_{t-1}
            dm = fma(nth, dm, -1.0);
            db = fma(nth, db, -yp);
            const double c = fma(nbeta, yp, yt - mu);
            err = fma(nth, err, c);
            ss = fma(err, err, ss);
            gm = fma(err, dm, gm);
            gb = fma(err, db, gb);
            gt = fma(err, dt, gt);
        };
        auto chunk = [&](double cy, const double (&Y)[8]) {   // 8 steps; cy = the y before Y[0]
            step(cy, Y[0]);
#pragma unroll
            for (int k = 1; k < 8; ++k) step(Y[k - 1], Y[k]);
        };
        // two register sets take turns: A = y[t .. t+7], B = y[t+8 .. t+15] (0-based), c0 = y[t-1]
        double A[8], B[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) A[k] = y[1 + k];
        int t = 1;
        for (; t + 16 <= T; t += 16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) B[k] = y[t + 8 + k];
            __builtin_amdgcn_sched_barrier(0);       // the scalar load stays in front of the chunk it overlaps
            chunk(c0, A);
            c0 = A[7];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) A[k] = y[t + 16 + k];
            __builtin_amdgcn_sched_barrier(0);
            chunk(c0, B);
            c0 = B[7];
        }
        int rem = T - t;                             // 0..15 steps left; A holds the first 8 of their y
        if (rem >= 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) B[k] = y[t + 8 + k];
            __builtin_amdgcn_sched_barrier(0);
            chunk(c0, A);
            c0 = A[7];
#pragma unroll
            for (int k = 0; k < 8; ++k) A[k] = B[k];
            rem -= 8;
        }
        if (rem > 0) step(c0, A[0]);
#pragma unroll
        for (int k = 1; k < 7; ++k)
            if (k < rem) step(A[k - 1], A[k]);
        // arma.stan:20-23 priors, + s for the Jacobian of sigma = exp(s)
        const double e2s = exp_fast(2.0 * s);  // sigma^2
        const double w = rcp_nr(e2s);          // 1 / sigma^2
        const double z2 = e2s * 0.16;          // (sigma / 2.5)^2
        double inv1pz;
        const double l1p = log1p_pos(z2, inv1pz);
        lpri = (-0.5 * kLog2Pi - 2.302585092994045684 - 0.005 * mu * mu)
             + (-0.5 * kLog2Pi - 0.6931471805599453094 - 0.125 * beta * beta)
             + (-0.5 * kLog2Pi - 0.6931471805599453094 - 0.125 * theta * theta)
             + (-kLogPi - 0.9162907318741550651 - l1p)
             + s;
        gp[0] = -0.01 * mu;
        gp[1] = -0.25 * beta;
        gp[2] = -0.25 * theta;
        gp[3] = 1.0 - 2.0 * (z2 * inv1pz);
        // arma.stan:30 normal_lpdf(err | 0, sigma)
        const double Td = (double)T;
        llik = -0.5 * Td * kLog2Pi - Td * s - 0.5 * ss * w;
        const double nw = -w;
        gl[0] = nw * gm;
        gl[1] = nw * gb;
        gl[2] = nw * gt;
        gl[3] = ss * w - Td;
    }
};

// LDS pairs (16 B) per lane and overflow pairs per lane
__host__ __device__ constexpr int n3_lds_pairs(int D, int LC, int LF) {
    const int VH = n2_vp(D) / 2;
    return (2 * VH + 1) + 3 * VH + 4 + 2 + LC * (2 * VH + 1) + LF * 2 * VH;
}
__host__ __device__ constexpr int n3_ovf_pairs(int D, int LC, int LF) {
    const int VH = n2_vp(D) / 2;
    return (10 - LC) * (2 * VH + 1) + (10 - LF) * 2 * VH;
}

template <class Model, bool TAPE, int LC, int LF>
__global__ void __launch_bounds__(kN3Block) nuts3_kernel(Nuts2Args a) {
    constexpr int D = Model::D, VP = n2_vp(D), VH = VP / 2;
    constexpr int INSZ = n2_in_doubles(D), OUTSZ = n2_out_doubles(D);
    // ---- lane-private LDS, in 16-byte pairs: pair P of lane l sits at lds3[P * 64 + l] ----------
    constexpr int REC = 0, RLP = 2 * VH;                       // x'(VH) r'(VH) (lpri1, llik1)
    constexpr int EDGE = RLP + 1;                              // x, r, grad of the edge that is not moving
    constexpr int RING = EDGE + 3 * VH;                        // 8 uniforms
    constexpr int NST = RING + 4;                              // n' of the parked halves, u16 x 10
    constexpr int CAND0 = NST + 2, CREC = 2 * VH + 1;          // candidate: x(VH) r(VH) (lpri, llik)
    constexpr int FIRST0 = CAND0 + LC * CREC, FREC = 2 * VH;   // first leaf of a sub-tree: x(VH) r(VH)
    constexpr int OCAND0 = 0, OFIRST0 = (10 - LC) * CREC, OVFP = n3_ovf_pairs(D, LC, LF);
    static_assert(FIRST0 + LF * FREC == n3_lds_pairs(D, LC, LF), "LDS layout");
    static_assert(LC >= 0 && LC <= 10 && LF >= 0 && LF <= 10, "levels");
    enum { INIT = 1, LEAF = 2, DONE = 3 };

    using d2 = double __attribute__((ext_vector_type(2)));
    using gptr2 = __attribute__((address_space(1))) d2*;
    using gcptr2 = const __attribute__((address_space(1))) d2*;
    using gcptr = const __attribute__((address_space(1))) double*;
    extern __shared__ d2 lds3[];
    const int lane = (int)threadIdx.x;
    d2* const L = lds3 + lane;
    gptr2 const O = (gptr2)a.ovf + ((int64_t)blockIdx.x * OVFP) * 64 + lane;

    Model model;
    model.init(a.mdata);
    const int64_t N = a.N;
    const double eps = a.eps, phi = a.phi;
    const int64_t p = (int64_t)blockIdx.x * kN3Block + lane;
    const bool live = p < N;
    const int64_t pc = live ? p : N - 1;   // idle lanes read (never write) the last particle's records

    // ---- vector moves: VH 16-byte accesses; `ptr` points at the lane's pair 0 of the record ------
    auto st_vec = [&](auto ptr, const double (&v)[D]) {
#pragma unroll
        for (int k = 0; k < VH; ++k) {
            d2 t;
            t.x = v[2 * k];
            t.y = (2 * k + 1 < D) ? v[2 * k + 1 < D ? 2 * k + 1 : 0] : 0.0;
            ptr[k * 64] = t;
        }
    };
    auto ld_vec = [&](auto ptr, double (&v)[D]) {
#pragma unroll
        for (int k = 0; k < VH; ++k) {
            const d2 t = ptr[k * 64];
            v[2 * k] = t.x;
            if (2 * k + 1 < D) v[2 * k + 1 < D ? 2 * k + 1 : 0] = t.y;
        }
    };
    // the parked candidate of level m / the first leaf of level s (1-based), wherever they live:
    // f(pointer) is instantiated once for LDS and once for the overflow area
    auto with_cand = [&](int m, auto&& f) {
        if (LC == 10 || m < LC) f(L + (CAND0 + m * CREC) * 64);
        else f(O + (OCAND0 + (m - LC) * CREC) * 64);
    };
    auto with_first = [&](int s, auto&& f) {
        if (LF == 10 || s - 1 < LF) f(L + (FIRST0 + (s - 1) * FREC) * 64);
        else f(O + (OFIRST0 + (s - 1 - LF) * FREC) * 64);
    };
    auto nst_ptr = [&](int m) -> unsigned short* {
        return reinterpret_cast<unsigned short*>(L + (NST + (m >> 3)) * 64) + (m & 7);
    };
    auto is_uturn = [](double A, double B, int dir) {
        // dir > 0: minus = other, plus = current: (A < 0) || (B < 0); dir < 0: dx, roles negate
        return dir > 0 ? ((A < 0.0) || (B < 0.0)) : ((B > 0.0) || (A > 0.0));
    };

    // ---- input / output records (layouts of smcn_nuts2.hpp: prep and post kernels are shared) ----
    gcptr2 const in2 = (gcptr2)a.in;
    gptr2 const out2 = (gptr2)a.out;
    auto in_rec = [&](int bb) -> gcptr2 { return in2 + ((int64_t)bb * N + pc) * (INSZ / 2); };

    // ---- per-lane state -------------------------------------------------------------------------
    int phase = live ? INIT : DONE;
    double x[D], r[D], g[D];
    double logu = 0.0, lpri0 = 0.0, llik0 = 0.0;
    int j = 0, i = 0, dir = 0, n = 1, nleap = 0, b = 0;
    uint32_t q = 1, qfill = 0;
    int64_t toff = 0, tlen = 0;
    bool overflow = false;
    d2 pre_r[VH], pre_e;          // the next transition's momentum and slice exponential, in flight

    auto request = [&](int bb) {
#pragma unroll
        for (int k = 0; k < VH; ++k) pre_r[k] = in_rec(bb)[VH + k];
        pre_e = in_rec(bb)[2 * VH];
    };
    auto begin_tree = [&](int bb) {   // r, e0 from the prefetched record; x is already in place
#pragma unroll
        for (int k = 0; k < VH; ++k) {
            r[2 * k] = pre_r[k].x;
            if (2 * k + 1 < D) r[2 * k + 1 < D ? 2 * k + 1 : 0] = pre_r[k].y;
        }
        logu = pre_e.x;               // raw; becomes H0 - e0 after the first evaluation
        q = 1; qfill = 0; overflow = false; nleap = 0;
        b = bb;
        if (bb + 1 < a.B) request(bb + 1);
        phase = INIT;
    };
    auto refill = [&]() {             // draws qfill, qfill + 1 of this particle's NUTS stream
        const u32x4 o = philox4x32_10({qfill >> 1, (uint32_t)(a.particle_base + p), a.iter + (uint32_t)b, kStreamNuts},
                                      (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
        d2 t;
        t.x = u53(o.a, o.b);
        t.y = u53(o.c, o.d);
        L[(RING + ((qfill >> 1) & 3u)) * 64] = t;
        qfill += 2u;
    };
    auto draw = [&]() -> double {
        double v;
        if constexpr (TAPE) {
            if ((int64_t)q < tlen) v = ((gcptr)a.tape)[toff + q];
            else { v = 0.5; overflow = true; }
        } else {
            const double* ring = reinterpret_cast<const double*>(L + (RING + ((q >> 1) & 3u)) * 64);
            v = ring[q & 1u];
        }
        ++q;
        return v;
    };

    {   // x0 and the first record
        const gcptr2 rec = in_rec(0);
#pragma unroll
        for (int k = 0; k < VH; ++k) {
            const d2 t = rec[k];
            x[2 * k] = t.x;
            if (2 * k + 1 < D) x[2 * k + 1 < D ? 2 * k + 1 : 0] = t.y;
        }
#pragma unroll
        for (int k = 0; k < D; ++k) { r[k] = 0.0; g[k] = 0.0; }
        request(0);
        if constexpr (TAPE) {
            toff = ((const __attribute__((address_space(1))) int64_t*)a.tape_off)[pc];
            tlen = ((const __attribute__((address_space(1))) int64_t*)a.tape_off)[pc + 1] - toff;
        }
        begin_tree(0);
        if (!live) phase = DONE;
    }

    for (;;) {
        const bool act = phase != DONE;
        // ---- uniforms: at least min(draws this leaf can consume, 7) in the ring -----------------
        if constexpr (!TAPE) {
            // leaf i of doubling j merges its trailing-one levels, then parks -- or, if that reaches
            // level j, draws the top-level accept and the next direction
            int need = 1;
            if (phase == LEAF) {
                const int t1 = __builtin_ctz(~(unsigned)i);
                need = (t1 >= j) ? j + 2 : t1;
            }
            need = need > 7 ? 7 : need;
            for (;;) {
                const int avail = (int)(qfill - q);
                if (__ballot(act && avail < need) == 0ull) break;
                if (act && avail <= 6) refill();
            }
        }

        // ---- leapfrog, first half (nuts.py:169-170) -------------------------------------------
        const double e = dir * eps, h = dir * eps / 2;
        if (phase == LEAF) {
#pragma unroll
            for (int k = 0; k < D; ++k) r[k] = r[k] + h * g[k];
#pragma unroll
            for (int k = 0; k < D; ++k) x[k] = x[k] + e * r[k];
        }
        double lpri = 0.0, llik = 0.0, gp[D], gl[D];
#pragma unroll
        for (int k = 0; k < D; ++k) { gp[k] = 0.0; gl[k] = 0.0; }
        if (act) model.eval(x, lpri, llik, gp, gl);
        double lp = lpri + phi * llik;
        const bool bad = !finite_d(lp);   // bridgestan.py:47-49,79-80
        lp = bad ? -kInf : lp;
#pragma unroll
        for (int k = 0; k < D; ++k) g[k] = bad ? -kInf : fma(phi, gl[k], gp[k]);

        bool start_doubling = false;
        if (phase == LEAF) {
            // ---- second half kick (nuts.py:173), leaf tests (:123-125) ----------------------------
            double kin = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) r[k] = r[k] + h * g[k];
#pragma unroll
            for (int k = 0; k < D; ++k) kin = fma(r[k], r[k], kin);
            ++nleap;
            const double joint = lp - 0.5 * kin;
            int nsub = (logu < joint) ? 1 : 0;
            bool ssub = (logu - a.delta_max) >= joint;
            // the sub-tree's candidate is kept BY REFERENCE: -1 = this leaf (x, r, lpri, llik in
            // registers), m >= 0 = the record parked in cand[m]; it is only copied when parked one
            // level up or accepted at the top
            int csrc = -1;
            if (j > 0 && (i & 1) == 0) {
                const int s = (i == 0) ? j : (__ffs(i) - 1);
                with_first(s, [&](auto fp) { st_vec(fp, x); st_vec(fp + VH * 64, r); });
            }
            auto cand_value = [&](double (&cx)[D], double (&cr)[D], d2& cl) {   // the candidate csrc refers to
                if (csrc < 0) {
#pragma unroll
                    for (int k = 0; k < D; ++k) { cx[k] = x[k]; cr[k] = r[k]; }
                    cl.x = lpri; cl.y = llik;
                } else {
                    with_cand(csrc, [&](auto cp) { ld_vec(cp, cx); ld_vec(cp + VH * 64, cr); cl = cp[2 * VH * 64]; });
                }
            };
            // ---- merges (nuts.py:134-148), the top level (:99-105) being level j ------------------
            bool done = false, stop = false;
            int m = 0;
            for (;;) {
                if (ssub) {
                    // unwinding: every ancestor whose SECOND half stopped still draws (:142)
                    q += (uint32_t)__popc((unsigned)(i >> m) & ((1u << (j - m)) - 1u));
                    done = true; stop = true;
                    break;
                }
                if constexpr (!TAPE) {   // a level beyond what the ring held at the top of the iteration
                    if (m >= 5 && q == qfill) refill();
                }
                if (m == j) {
                    // top level: accept with prob min(1, n'/n) (:99), U-turn on the outer edges (:105)
                    const double u = draw();
                    if (nsub >= n || fma(u, (double)n, -(double)nsub) < 0.0) {
                        double cx[D], cr[D];
                        d2 cl;
                        cand_value(cx, cr, cl);
                        st_vec(L + REC * 64, cx); st_vec(L + (REC + VH) * 64, cr);
                        L[(REC + RLP) * 64] = cl;
                    }
                    double xo[D], ro[D], A = 0.0, B = 0.0;
                    ld_vec(L + EDGE * 64, xo); ld_vec(L + (EDGE + VH) * 64, ro);
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        const double d = x[k] - xo[k];
                        A = fma(d, ro[k], A);
                        B = fma(d, r[k], B);
                    }
                    stop = is_uturn(A, B, dir);
                    done = true;
                    break;
                }
                if (((i >> m) & 1) == 0) {   // first half of level m+1: park it
                    double cx[D], cr[D];
                    d2 cl;
                    cand_value(cx, cr, cl);
                    with_cand(m, [&](auto crec) { st_vec(crec, cx); st_vec(crec + VH * 64, cr); crec[2 * VH * 64] = cl; });
                    *nst_ptr(m) = (unsigned short)nsub;
                    break;
                }
                const int i0 = (i >> (m + 1)) << (m + 1);
                const int s = (i0 == 0) ? j : (__ffs(i0) - 1);
                const double u = draw();     // :142, always
                double fx[D], fr[D];
                const int n1 = (int)*nst_ptr(m);
                with_first(s, [&](auto fp) { ld_vec(fp, fx); ld_vec(fp + VH * 64, fr); });
                const int den = (n1 + nsub) > 1 ? (n1 + nsub) : 1;
                const bool keep = !(fma(u, (double)den, -(double)nsub) < 0.0);   // keep the first half's candidate
                csrc = keep ? m : csrc;
                nsub += n1;                  // :146
                double A = 0.0, B = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double d = x[k] - fx[k];
                    A = fma(d, fr[k], A);
                    B = fma(d, r[k], B);
                }
                ssub = is_uturn(A, B, dir);  // :148
                ++m;
            }
            if (!done) {
                ++i;
            } else {
                n += nsub;                   // :103  (unused after a stop)
                ++j;
                if (stop || j > a.max_depth) {   // :89,109 -> emit the output record, start the next transition
                    const gptr2 orec = out2 + ((int64_t)b * N + p) * (OUTSZ / 2);
#pragma unroll
                    for (int k = 0; k < 2 * VH + 1; ++k) orec[k] = L[(REC + k) * 64];
                    d2 t;
                    t.x = lpri0; t.y = llik0;
                    orec[2 * VH + 1] = t;
                    const unsigned long long s0 = (unsigned long long)(unsigned)nleap | ((unsigned long long)(unsigned)j << 32);
                    const unsigned long long s1 = (unsigned long long)q | ((unsigned long long)(overflow ? 1u : 0u) << 32);
                    t.x = __longlong_as_double((long long)s0);
                    t.y = __longlong_as_double((long long)s1);
                    orec[2 * VH + 2] = t;
                    if (b + 1 < a.B) {
                        ld_vec(L + REC * 64, x);   // continue from the sample just drawn
                        begin_tree(b + 1);
                    } else {
                        phase = DONE;
                    }
                } else {
                    start_doubling = true;
                }
            }
        } else if (phase == INIT) {
            // ---- nuts.py:66-87 ----------------------------------------------------------------
            double kin = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) kin = fma(r[k], r[k], kin);
            logu = (lp - 0.5 * kin) - logu;      // H0 - Exp(1)
            lpri0 = lpri; llik0 = llik;           // the record's start density
            st_vec(L + REC * 64, x); st_vec(L + (REC + VH) * 64, r);
            { d2 t; t.x = lpri; t.y = llik; L[(REC + RLP) * 64] = t; }
            j = 0; n = 1;
            dir = 0;                              // both edges are (x0, r0, g0)
            start_doubling = true;
            phase = LEAF;
        }
        if (start_doubling) {
            // ---- nuts.py:91: direction; the moving state becomes that edge --------------------------
            const int nd = (draw() < 0.5) ? 1 : -1;
            if (dir == 0) {
                st_vec(L + EDGE * 64, x); st_vec(L + (EDGE + VH) * 64, r); st_vec(L + (EDGE + 2 * VH) * 64, g);
            } else if (nd != dir) {           // the moving edge and the parked one trade places
                double ox[D], orr[D], og[D];
                ld_vec(L + EDGE * 64, ox); ld_vec(L + (EDGE + VH) * 64, orr); ld_vec(L + (EDGE + 2 * VH) * 64, og);
                st_vec(L + EDGE * 64, x); st_vec(L + (EDGE + VH) * 64, r); st_vec(L + (EDGE + 2 * VH) * 64, g);
#pragma unroll
                for (int k = 0; k < D; ++k) { x[k] = ox[k]; r[k] = orr[k]; g[k] = og[k]; }
            }
            dir = nd;
            i = 0;
        }
        if (__ballot(phase != DONE) == 0ull) break;
    }
}

}  // namespace smcn
