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
// There is no work queue: lane l of block b owns particle 64 b + l for all B fused transitions (QUEUE = false; with fewer
// lanes than particles -- QUEUE = true -- a wavefront owns a run of particles and hands their segments out to its lanes).
#pragma once
#include "smcn_nuts2.hpp"
#include <type_traits>

namespace smcn {

constexpr int kN3Block = 64;   // one wavefront per block: no barrier, no coupling between waves
constexpr int kN3ReadyWords = 64;   // lane queue: 64-bit ready words of a wavefront, one per lane

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
// The fp64 constants of ArmaLaneModel::finish (exp_fast, log_ge1 of smcn_device.hpp, the same values in the same
// order of use).  As literals they become ~30 scalar register PAIRS that the compiler keeps alive across the whole leaf
// loop -- and, with 100 scalar registers taken, spills to vector lanes and back around every evaluation; read by scalar
// loads where finish starts (through a laundered pointer, so that the loads stay there) they are live for 700 cycles of 14 000.
__constant__ double kArmaFinishTab[24] = {
    1.4426950408889634074, 6.93147180369123816490e-01, 1.90821492927058770002e-10,          // 1 / ln 2, ln 2 hi, lo
    2.08767569878680989792e-09, 2.50521083854417187751e-08, 2.75573192239858906526e-07,     // 1 / 12! ..
    2.75573192239858906526e-06, 2.48015873015873015873e-05, 1.98412698412698412698e-04,
    1.38888888888888888889e-03, 8.33333333333333333333e-03, 4.16666666666666666667e-02,
    1.66666666666666666667e-01,                                                             // .. 1 / 3!
    0.70710678118654752440,                                                                 // sqrt(1/2)
    9.52380952380952380952e-02, 1.05263157894736842105e-01, 1.17647058823529411765e-01,     // 2/21, 2/19, ..
    1.33333333333333333333e-01, 1.53846153846153846154e-01, 1.81818181818181818182e-01,
    2.22222222222222222222e-01, 2.85714285714285714286e-01, 4.0e-01, 6.66666666666666666667e-01};

struct ArmaLaneModel {
    static constexpr int D = 4;
    static constexpr bool HAS_WIDE = true;     // recur_wide<A>: A lanes per particle for a wavefront's last stragglers
    using cptr = const __attribute__((address_space(4))) double*;
    int T;
    cptr y;

    __device__ __forceinline__ void init(const double* md) {
        T = __builtin_amdgcn_readfirstlane((int)((cptr)md)[0]);   // (the conversion is a vector instruction)
        y = (cptr)md + 1;
    }

    __device__ __forceinline__ void eval(const double (&x)[4], double& lpri, double& llik, double (&gp)[4],
                                         double (&gl)[4]) const {
        double ss, gm, gb, gt;
        recur(x, ss, gm, gb, gt);
        finish(x, ss, gm, gb, gt, lpri, llik, gp, gl);
    }

    // The T-step recurrence: ss = sum err_t^2 and gm, gb, gt = sum err_t * d err_t / d(mu, beta, theta).
    __device__ __forceinline__ void recur(const double (&x)[4], double& ss_o, double& gm_o, double& gb_o,
                                          double& gt_o) const {
        const double mu = x[0], beta = x[1], theta = x[2];
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
        // every scalar load so far has landed before the loop is entered: otherwise the compiler's conservative
        // merge of the pre-header's pending loads makes it wait INSIDE the loop, 40 instructions after a load
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
        // Scalar loads return out of order, so a wait is always for ALL of them: each load is therefore issued
        // right AFTER the wait for the previous one (behind the first step of the chunk that consumes it) and has
        // the other seven steps of that chunk plus the first of the next to land.
        auto round16 = [&](int tt) __attribute__((always_inline)) {   // 16 steps from A = y[tt ..]; leaves A = y[tt + 16 ..]
            step(c0, A[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) B[k] = y[tt + 8 + k];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 1; k < 8; ++k) step(A[k - 1], A[k]);
            c0 = A[7];
            __builtin_amdgcn_sched_barrier(0);       // (nothing of the next chunk moves up in front of its wait)
            step(c0, B[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) A[k] = y[tt + 16 + k];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 1; k < 8; ++k) step(B[k - 1], B[k]);
            c0 = B[7];
            __builtin_amdgcn_sched_barrier(0);
        };
        // four rounds per trip: a taken branch costs a lone wavefront ~30 cycles of instruction refetch
        for (; t + 64 <= T; t += 64) {
            round16(t);
            round16(t + 16);
            round16(t + 32);
            round16(t + 48);
        }
        if (t + 32 <= T) {
            round16(t);
            round16(t + 16);
            t += 32;
        }
        if (t + 16 <= T) {
            round16(t);
            t += 16;
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
        // 0..7 steps left, taken 4, 2, 1 at a time (three scalar branches, not seven: each one a lone wavefront
        // takes costs it ~20 cycles); A is shifted down behind each group
        if (rem >= 4) {
            step(c0, A[0]); step(A[0], A[1]); step(A[1], A[2]); step(A[2], A[3]);
            c0 = A[3];
#pragma unroll
            for (int k = 0; k < 4; ++k) A[k] = A[k + 4];
            rem -= 4;
        }
        if (rem >= 2) {
            step(c0, A[0]); step(A[0], A[1]);
            c0 = A[1];
            A[0] = A[2];
            rem -= 2;
        }
        if (rem >= 1) step(c0, A[0]);
        ss_o = ss; gm_o = gm; gb_o = gb; gt_o = gt;
    }

    // Priors, Jacobian and the likelihood's closed part, from the recurrence's four sums.
    __device__ __forceinline__ void finish(const double (&x)[4], double ss, double gm, double gb, double gt, double& lpri,
                                           double& llik, double (&gp)[4], double (&gl)[4]) const {
        const double mu = x[0], beta = x[1], theta = x[2], s = x[3];
        cptr tab = (cptr)kArmaFinishTab;
        asm volatile("" : "+s"(tab));          // (the scalar loads below are not hoisted out of the leaf loop)
        double K[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) K[i] = tab[i];
        // arma.stan:20-23 priors, + s for the Jacobian of sigma = exp(s)
        // e^(2s): exp_fast of smcn_device.hpp with its constants from the table
        double e2s;                            // sigma^2
        {
            const double t = 2.0 * s;
            const double k = __builtin_rint(t * K[0]);
            double r = fma(-k, K[1], t);
            r = fma(-k, K[2], r);
            double p = K[3];
#pragma unroll
            for (int i = 4; i <= 12; ++i) p = fma(p, r, K[i]);
            p = fma(p, r, 0.5);
            p = fma(p, r, 1.0);
            p = fma(p, r, 1.0);
            e2s = ldexp(p, (int)k);
        }
        const double w = rcp_nr(e2s);          // 1 / sigma^2
        const double z2 = e2s * 0.16;          // (sigma / 2.5)^2
        // log1p_pos(z2) with 1 / (1 + z2): log_ge1 of smcn_device.hpp with its constants from the table
        double inv1pz, l1p;
        {
            const double u = 1.0 + z2;
            inv1pz = rcp_nr(u);
            int e = __builtin_amdgcn_frexp_exp(u);
            double m = __builtin_amdgcn_frexp_mant(u);
            const bool lo = m < K[13];
            m = lo ? m + m : m;
            e = lo ? e - 1 : e;
            const double f = (m - 1.0) * rcp_nr(m + 1.0);
            const double f2 = f * f;
            double p = K[14];
#pragma unroll
            for (int i = 15; i <= 23; ++i) p = fma(p, f2, K[i]);
            const double logm = fma(f * f2, p, f + f);
            const double ed = (double)e;
            const double lg = fma(ed, K[1], fma(ed, K[2], logm));
            l1p = fma(z2 - (u - 1.0), inv1pz, lg);
        }
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


    // ---- the same four sums with A lanes per particle -------------------------------------------------------------
    // The launch lasts as long as its longest chain of leaves, and towards its end a wavefront steps a handful of
    // stragglers while most lanes idle.  When at most 64 / A lanes are active, lanes [gA, (g+1)A) work for the g-th
    // active lane: the T - 1 steps are cut into A consecutive segments, one per lane.
    //   pass 1  every lane runs the STATE recurrences (err, d err) of its segment from a zero incoming state (segment 0
    //           from the true start): 6 instructions per step.  A segment's end state is affine in its incoming state,
    //               err' = P err + E,  dm' = P dm + M,  db' = P db + B,  dt' = P dt + Q err + Tt,
    //           with P = (-theta)^n, Q = -n (-theta)^(n-1) for its n steps;
    //   scan    Hillis-Steele composition of these maps over the A lanes (DPP shifts); segment 0 is a constant map, so
    //           after log2 A stages lane a holds the true end state of segment a, and its neighbour's is its start;
    //   pass 2  the full ten-instruction step from the true incoming state, sums per segment, butterfly over the group.
    // About (16 S + 150) instructions for S = ceil((T - 1) / A) steps instead of 10 T.  The sums are re-associated
    // (segment by segment), so the result differs from recur() by rounding (~1e-15 relative) -- see DESIGN.md 4.1 for
    // what that means for bit-for-bit comparisons between differently scheduled runs.
    static constexpr int YMAX = 384;       // series that fit the LDS copy the segments read (longer ones stay narrow)
    static constexpr int WIDE_MIN_T = 64;
    static constexpr int WIDE_MIN_T_ROWS = 130; // 32 / 64 lanes per series: every segment at least two steps
    static constexpr int YPAD = 8;         // doubles a segment's chunked reads may run past the series
    template <int A>
    static constexpr int xch_pairs() { return 2 * (64 / A); }
    // the value of lane a - K of the group.  A <= 16 (a DPP row / a quad): lanes a < K have absorbed segment 0 by then --
    // a constant map, whatever they compose it with -- so their fill does not matter.  A = 32, 64 (two / four rows):
    // the shift stays inside a row and a lane whose source would lie before its row receives `fill`, the component of
    // the IDENTITY map; the rows are joined afterwards (row_join).
    template <int CTRL, int ROWS>
    static __device__ __forceinline__ double dpp_fill(double v, double fill) {
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, CTRL, ROWS, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, CTRL, ROWS, 0xF, false);
        return __hiloint2double(hi, lo);
    }
    template <int A, int K>
    static __device__ __forceinline__ double seg_shift(double v, double fill) {
        static_assert(A == 64 || A == 32 || A == 16 || A == 8 || A == 4, "group widths: rows, a DPP row, half a row or a quad");
        if constexpr (A >= 32) return dpp_fill<0x110 + K, 0xF>(v, fill);     // row_shr:K
        else if constexpr (A >= 8) return dpp_mov<0x110 + K>(v);             // row_shr:K, zero fill (A = 8: what crosses
                                                                             // into a row's second half meets a constant map)
        else return dpp_mov<(K == 1) ? 0x90 : 0x44>(v);                      // quad_perm [0,0,1,2] / [0,1,0,1]
    }
    // the totals of the row before (lane 15 of it: row_bcast:15) for the rows of ROWS / of rows 0-1 (lane 31: row_bcast:31)
    template <int CTRL, int ROWS>
    static __device__ __forceinline__ double row_join(double v, double fill) { return dpp_fill<CTRL, ROWS>(v, fill); }
    // One pass over a lane's segment: `step(y_{t-1}, y_t)` for the S - 1 steps every lane of the wavefront takes, then
    // one more on the lanes with `extra`.  The series values come from LDS at the lane's own index, four steps per
    // read and one chunk AHEAD of their use (a lone wavefront cannot hide an LDS round trip behind another's work).
    // Yp points at the observation before the segment; reads run up to 7 doubles past a segment's end (the LDS copy
    // is padded).
    template <class F>
    static __device__ __forceinline__ void seg_walk(const double* Yp, int S, bool extra, F&& step) {
        const int nfull = (S - 1) >> 2, rem = (S - 1) & 3;        // wave-uniform
        double yp = Yp[0];
        double c0 = Yp[1], c1 = Yp[2], c2 = Yp[3], c3 = Yp[4];
        const double* q = Yp + 5;
        for (int c = 0; c < nfull; ++c) {
            const double n0 = q[0], n1 = q[1], n2 = q[2], n3 = q[3];   // the next chunk (or the tail's values)
            q += 4;
            step(yp, c0); step(c0, c1); step(c1, c2); step(c2, c3);
            yp = c3;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        }
        if (rem > 0) { step(yp, c0); yp = c0; c0 = c1; c1 = c2; c2 = c3; }
        if (rem > 1) { step(yp, c0); yp = c0; c0 = c1; c1 = c2; }
        if (rem > 2) { step(yp, c0); yp = c0; c0 = c1; }
        if (extra) step(yp, c0);
    }
    template <int A>
    __device__ __forceinline__ void recur_wide(const double (&x)[4], bool act, unsigned long long mask, const double* Yl,
                                               double __attribute__((ext_vector_type(2)))* XCH, int lane, double& ss_o,
                                               double& gm_o, double& gb_o, double& gt_o) const {
        using d2 = double __attribute__((ext_vector_type(2)));
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        if (act) {
            d2 t;
            t.x = x[0]; t.y = x[1];
            XCH[2 * rank] = t;
            t.x = x[2]; t.y = x[3];
            XCH[2 * rank + 1] = t;
        }
        wave_exchange_fence();
        const int g = lane / A, a = lane % A;
        const d2 q0 = XCH[2 * g], q1 = XCH[2 * g + 1];
        const double mu = q0.x, beta = q0.y, theta = q1.x;
        const double nth = -theta, nbeta = -beta;
        // segments: n = T - 1 steps (observations 1 .. T-1, 0-based); the first R segments take S steps, the others S - 1
        const int n = T - 1, S = (n + A - 1) / A, R = n - A * (S - 1);
        const bool extra = a < R;
        const int i0 = 1 + a * (S - 1) + (a < R ? a : R);
        const bool first = a == 0;
        // observation 0 (arma.stan:25: nu_1 = mu + beta mu): the start of segment 0
        const double y0 = Yl[0];
        const double e_init = fma(nbeta, mu, y0 - mu), dm_init = -(1.0 + beta), db_init = -mu;
        double e = first ? e_init : 0.0, dm = first ? dm_init : 0.0, db = first ? db_init : 0.0, dt = 0.0;
        auto state_step = [&](double yp, double yt) __attribute__((always_inline)) {
            dt = fma(nth, dt, -e);
            dm = fma(nth, dm, -1.0);
            db = fma(nth, db, -yp);
            e = fma(nth, e, fma(nbeta, yp, yt - mu));
        };
        seg_walk(Yl + (i0 - 1), S, extra, state_step);
        // (-theta)^(S-2) by squaring (S >= 2: T >= WIDE_MIN_T), then the segment's P and Q
        double pm = 1.0;
        {
            double base = nth;
            for (int m = S - 2; m > 0; m >>= 1) {
                pm = (m & 1) ? pm * base : pm;
                base = base * base;
            }
        }
        const double p1 = pm * nth, p2 = p1 * nth;          // (-theta)^(S-1), (-theta)^S
        double P = first ? 0.0 : (extra ? p2 : p1);
        double Q = first ? 0.0 : (extra ? -(double)S * p1 : -(double)(S - 1) * pm);
        double E = e, M = dm, Bv = db, Tt = dt;
        auto stage = [&](double Pl, double Ql, double El, double Ml, double Bl, double Tl) __attribute__((always_inline)) {
            E = fma(P, El, E);
            M = fma(P, Ml, M);
            Bv = fma(P, Bl, Bv);
            Tt = fma(P, Tl, fma(Q, El, Tt));
            Q = fma(P, Ql, Q * Pl);
            P = P * Pl;
        };
#define SMCN_WIDE_STAGE(K) stage(seg_shift<A, K>(P, 1.0), seg_shift<A, K>(Q, 0.0), seg_shift<A, K>(E, 0.0), seg_shift<A, K>(M, 0.0), seg_shift<A, K>(Bv, 0.0), seg_shift<A, K>(Tt, 0.0))
#define SMCN_WIDE_JOIN(C, R) stage(row_join<C, R>(P, 1.0), row_join<C, R>(Q, 0.0), row_join<C, R>(E, 0.0), row_join<C, R>(M, 0.0), row_join<C, R>(Bv, 0.0), row_join<C, R>(Tt, 0.0))
        SMCN_WIDE_STAGE(1);
        SMCN_WIDE_STAGE(2);
        if constexpr (A >= 8) SMCN_WIDE_STAGE(4);
        if constexpr (A >= 16) SMCN_WIDE_STAGE(8);
        if constexpr (A >= 32) SMCN_WIDE_JOIN(0x142, 0xA);   // rows 1, 3 take in rows 0, 2 (A = 32: the two groups' second rows)
        if constexpr (A == 64) SMCN_WIDE_JOIN(0x143, 0xC);   // rows 2, 3 take in rows 0-1
#undef SMCN_WIDE_JOIN
#undef SMCN_WIDE_STAGE
        // the neighbour's end state is this segment's start
        {
            auto prev = [](double v) __attribute__((always_inline)) {
                if constexpr (A >= 32) return dpp_mov<0x138>(v);     // wave_shr:1 (crosses rows)
                else return seg_shift<A, 1>(v, 0.0);
            };
            const double ei = prev(E), mi = prev(M), bi = prev(Bv), ti = prev(Tt);
            e = first ? e_init : ei;
            dm = first ? dm_init : mi;
            db = first ? db_init : bi;
            dt = first ? 0.0 : ti;
        }
        double ss = first ? e * e : 0.0, gm = first ? e * dm : 0.0, gb = first ? e * db : 0.0, gt = 0.0;
        auto full_step = [&](double yp, double yt) __attribute__((always_inline)) {
            dt = fma(nth, dt, -e);
            dm = fma(nth, dm, -1.0);
            db = fma(nth, db, -yp);
            e = fma(nth, e, fma(nbeta, yp, yt - mu));
            ss = fma(e, e, ss);
            gm = fma(e, dm, gm);
            gb = fma(e, db, gb);
            gt = fma(e, dt, gt);
        };
        seg_walk(Yl + (i0 - 1), S, extra, full_step);
        if constexpr (A == 64) wave_sum4(ss, gm, gb, gt, ss, gm, gb, gt);   // the four sums in one butterfly
        else { ss = group_sum<A>(ss); gm = group_sum<A>(gm); gb = group_sum<A>(gb); gt = group_sum<A>(gt); }
        if (first) {                 // (every lane has read its inputs: a wavefront executes in order)
            d2 t;
            t.x = ss; t.y = gm;
            XCH[2 * g] = t;
            t.x = gb; t.y = gt;
            XCH[2 * g + 1] = t;
        }
        wave_exchange_fence();
        if (act) {
            const d2 r0 = XCH[2 * rank], r1 = XCH[2 * rank + 1];
            ss_o = r0.x; gm_o = r0.y; gb_o = r1.x; gt_o = r1.y;
        }
    }
};

// Batched value + gradient of a LaneModel, one thread per row (smcn_target_eval, initial weights, tempering
// parts): x element (row i, coordinate c) at x[i*rs + c*cs]; outputs may be null.
template <class Model>
__global__ void __launch_bounds__(256) lane_eval_kernel(const double* mdata, const double* x, int64_t M, int64_t rs,
                                                        int64_t cs, double phi, double* logp, double* grad, int64_t grs,
                                                        int64_t gcs, double* lpri_o, double* llik_o) {
    constexpr int D = Model::D;
    Model model;
    model.init(mdata);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
        double xv[D], lpri, llik, gp[D], gl[D];
#pragma unroll
        for (int k = 0; k < D; ++k) xv[k] = x[i * rs + k * cs];
        model.eval(xv, lpri, llik, gp, gl);
        const double lp0 = lpri + phi * llik;
        const bool bad = !finite_d(lp0);
        if (logp) logp[i] = bad ? -kInf : lp0;
        if (lpri_o) lpri_o[i] = lpri;
        if (llik_o) llik_o[i] = llik;
        if (grad) {
#pragma unroll
            for (int k = 0; k < D; ++k) grad[i * grs + k * gcs] = bad ? -kInf : fma(phi, gl[k], gp[k]);
        }
    }
}

// Selftest of the wide evaluation (smcn_selftest_wide): block b evaluates rows [b G, (b+1) G), G = 64 / A, each owned by
// a lane scattered over the wavefront (so that ranks and lanes differ); out[row] = the four sums of recur(), then those of
// recur_wide<A>().
template <class Model, int A>
__global__ void __launch_bounds__(64) selftest_wide_kernel(const double* mdata, const double* x, int64_t M, double* out) {
    using d2 = double __attribute__((ext_vector_type(2)));
    extern __shared__ d2 lds_sw[];
    Model model;
    model.init(mdata);
    double* const Yl = reinterpret_cast<double*>(lds_sw);
    d2* const XCH = reinterpret_cast<d2*>(Yl + Model::YMAX + Model::YPAD);
    const int lane = (int)threadIdx.x;
    for (int i = lane; i < model.T + Model::YPAD && i < Model::YMAX + Model::YPAD; i += 64) Yl[i] = i < model.T ? mdata[1 + i] : 0.0;
    wave_exchange_fence();
    constexpr int G = 64 / A;
    const int j = lane / A;                                  // the row (within the block) this lane may own
    const bool owner = (lane % A) == ((5 * j + 3) % A);
    const int64_t row = (int64_t)blockIdx.x * G + (G - 1 - j);   // ranks run against the rows
    const bool act = owner && row < M;
    double xv[4] = {0.0, 0.0, 0.0, 0.0};
    if (act)
        for (int c = 0; c < 4; ++c) xv[c] = x[row * 4 + c];
    double n0 = 0, n1 = 0, n2 = 0, n3 = 0, w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    if (act) model.recur(xv, n0, n1, n2, n3);
    model.template recur_wide<A>(xv, act, __ballot(act), Yl, XCH, lane, w0, w1, w2, w3);
    if (act) {
        double* o = out + row * 8;
        o[0] = n0; o[1] = n1; o[2] = n2; o[3] = n3; o[4] = w0; o[5] = w1; o[6] = w2; o[7] = w3;
    }
}

// Where the per-lane tree state lives (one wavefront per SIMD: 512 VGPRs and 640 B of LDS per lane):
//   registers : moving state (x, r, grad), the parked edge, the accepted sample, tree-stack levels
//               0-1 (candidates 0, 1 and first leaves of levels 1, 2), 4 prefetched uniforms;
//   LDS       : 8-entry ring of uniforms, n' of the parked halves of levels >= 2 (u16), candidates of
//               levels 2 .. 2+LC-1 and first leaves of levels 3 .. 3+LF-1, lane-private, laid out
//               [16-byte pair][lane] (conflict-free b128 accesses at any mix of levels);
//   global    : deeper levels ([pair][lane] per wavefront); with LC = 3, LF = 3 only trees of more
//               than 32 leaves ever touch it.
// The first leaf of a sub-tree is stored once per level it opens (F[l], l = 1 .. ctz(i)), so that the
// merge of level m reads F[m+1] at a fixed place.
__host__ __device__ constexpr int n3_lds_pairs(int D, int LC, int LF) {
    const int VH = n2_vp(D) / 2;
    return 4 + 2 + (VH + 1) + LC * (2 * VH + 1) + LF * 2 * VH;
}
__host__ __device__ constexpr int n3_ovf_pairs(int D, int LC, int LF) {
    const int VH = n2_vp(D) / 2;
    return (8 - LC) * (2 * VH + 1) + (8 - LF) * 2 * VH;
}

// bytes of dynamic LDS per block: the lane-private pairs, then (models with a wide evaluation) the series copy and the
// exchange slots of the lane groups
template <class Model>
__host__ __device__ constexpr int n3_lds_bytes(int LC, int LF) {
    int b = 16 * kN3Block * n3_lds_pairs(Model::D, LC, LF);
    if constexpr (Model::HAS_WIDE) b += 8 * (Model::YMAX + Model::YPAD) + 16 * Model::template xch_pairs<4>();
    return b;
}

__device__ __forceinline__ bool compact_mode(const Nuts2Args& a) { return a.logw0 != nullptr; }

// QUEUE: the grid holds fewer lanes than there are particles (populations beyond one wavefront per SIMD, or a cap set
// with smcn_set_lane_grid).  Every wavefront OWNS a contiguous run of particles -- N / waves of them, the first N % waves
// wavefronts one more -- and its 64 lanes work through them: a job is a SEGMENT of a particle's block of B transitions
// (a.seg_len of them, the last segment cut once more a.seg_tail before the end; the whole block without segments); a lane
// takes a READY job of its wavefront, the least advanced particle first.  Ready jobs are bits of 64-bit words
// [segment][particle / 64] that live one per LANE in a register: set for segment 0 of the particles beyond the first 64
// at the start, set for segment s + 1 when a lane ends segment s and leaves (x', running log-weight) in
// a.handover[particle][s].  Nothing is asked of another wavefront: no atomics, no waiting for another lane's tree (what is
// ready has been written by this wavefront, in program order).  With 128 particles or more a wavefront its work is
// their SUM (spread ~1 %) where a lane per particle made it the longest of 64 chains.  (Tried first, and measured worse
// -- profiles/r04_lane_schedules.md: tickets from one global counter -- ~12 ns per atomic on one address --, from 8 and 64
// counters with polled hand-over slots, and global ready queues; every form that let a lane idle an odd number of loop
// iterations between two jobs made every iteration of its wavefront ~15 % dearer: see have_next / own_wait below.)
// QUEUE = false is the kernel without any of it (a lane per particle, as before).
template <class Model, bool TAPE, int LC, int LF, bool QUEUE = false, int WPE = 1>
__global__ void __launch_bounds__(kN3Block) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) nuts3_kernel(Nuts2Args a) {
    constexpr int D = Model::D, VP = n2_vp(D), VH = VP / 2;
    constexpr int INSZ = n2_in_doubles(D), OUTSZ = n2_out_doubles(D);
    // ---- lane-private LDS, in 16-byte pairs: pair P of lane l sits at lds3[P * 64 + l] ----------
    constexpr int RING = 0;                                    // 8 uniforms
    constexpr int NST = RING + 4;                              // n' of the parked halves, u16, index = level
    constexpr int PREF = NST + 2;                              // the next transition's momentum (VH) and slice exponential (1),
                                                               // loaded global -> LDS without passing through registers
    constexpr int CAND0 = PREF + VH + 1, CREC = 2 * VH + 1;    // candidate of level 2 + k: x(VH) r(VH) (lpri, llik)
    constexpr int FIRST0 = CAND0 + LC * CREC, FREC = 2 * VH;   // first leaf of level 3 + k: x(VH) r(VH)
    constexpr int OCAND0 = 0, OFIRST0 = (8 - LC) * CREC, OVFP = n3_ovf_pairs(D, LC, LF);
    static_assert(FIRST0 + LF * FREC == n3_lds_pairs(D, LC, LF), "LDS layout");
    static_assert(LC >= 0 && LC <= 8 && LF >= 0 && LF <= 8, "levels");
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
    // wide evaluation (Model::recur_wide): the series in LDS, read by the lanes of a group at their own time index
    double* const Yl = reinterpret_cast<double*>(lds3 + n3_lds_pairs(D, LC, LF) * 64);
    d2* const XCH = reinterpret_cast<d2*>(Yl + (Model::HAS_WIDE ? Model::YMAX + Model::YPAD : 0));
    bool wide_ok = false, wide_rows = false;
    if constexpr (Model::HAS_WIDE) {
        wide_ok = (a.wide & 1) != 0 && model.T >= Model::WIDE_MIN_T && model.T <= Model::YMAX;
        wide_rows = wide_ok && model.T >= Model::WIDE_MIN_T_ROWS;
        if (wide_ok) {
            for (int i = lane; i < model.T + Model::YPAD; i += kN3Block) Yl[i] = i < model.T ? ((gcptr)a.mdata)[1 + i] : 0.0;
            wave_exchange_fence();
        }
    }
    const int64_t N = a.N;
    const double eps = a.eps, phi = a.phi;
    // QUEUE: this wavefront's particles [w_start, w_start + w_count)
    const unsigned int w_quot = (unsigned int)N / gridDim.x, w_rem = (unsigned int)N % gridDim.x;
    const unsigned int w_count = QUEUE ? w_quot + (blockIdx.x < w_rem ? 1u : 0u) : (unsigned int)kN3Block;
    const int64_t w_start = QUEUE ? (int64_t)blockIdx.x * w_quot + (blockIdx.x < w_rem ? blockIdx.x : w_rem)
                                  : (int64_t)blockIdx.x * kN3Block;
    const int64_t p_first = w_start + lane;
    const bool live = QUEUE ? (unsigned int)lane < w_count : p_first < N;
    std::conditional_t<QUEUE, int64_t, const int64_t> p = p_first;   // (QUEUE: the lane's CURRENT particle)
    const int64_t pc = live ? p_first : N - 1;   // idle lanes read (never write) the last particle's records
    // QUEUE: the lane's NEXT job, taken from the wavefront's ready bits a whole tree before it is needed (when the lane starts
    // the last transition of its segment): its start state (x, running log-weight) waits in registers, its first record in
    // the prefetch slots, and the lane goes over to it in the iteration its segment ends -- no idle iteration, and the trees
    // of a wavefront's lanes stay in step (a tree takes 2^depth iterations, so lanes that never idle start theirs in
    // iterations of the same residue mod 2, 4, 8 and take the cheap and the expensive turns of the merge code together;
    // measured: hand-overs that shifted that residue cost every iteration of the wavefront ~15 %).  If nothing was ready
    // the lane goes on with its own particle's next segment, nothing handed over.  A tree that stopped early leaves its
    // lane out of step: it then sits out up to a.seg_align - 1 iterations before its next tree (own_wait), as does a lane
    // that had no job (measured at N = 131 072: +4 % for the former on top of +9 % for the latter).
    // n_ready: set ready bits (wave-uniform).
    bool have_next = false, own_wait = false;   // (own_wait: sits out iterations until its next tree would start in step)
    unsigned int i_next = 0u;        // (the next job: the wavefront's i-th particle, segment seg_next)
    int seg_next = 0;
    unsigned int n_ready = 0u;
    double lw_next = 0.0;
    double hx_next[D];
    int64_t toff_next = 0, tlen_next = 0;
    // segments (QUEUE with a.seg_len > 0): this lane runs transitions [b, b_end) of its particle's block
    // (segments of Bs transitions; the last of them is cut once more, a.seg_tail transitions before the block's end: the
    //  jobs a wavefront ends on are short, and so is the time its other lanes wait for them)
    const int Bs = (QUEUE && a.seg_len > 0) ? a.seg_len : a.B;
    const int n_main = (a.B + Bs - 1) / Bs, b_tail = (QUEUE && a.seg_tail > 0) ? a.B - a.seg_tail : a.B;
    const unsigned int nseg = (unsigned int)(n_main + (b_tail < a.B ? 1 : 0));
    auto seg_begin = [&](int sg) __attribute__((always_inline)) {    // first transition of segment sg (sg = nseg: the block's end)
        return sg < n_main ? sg * Bs : (sg == n_main ? b_tail : a.B);
    };
    int seg = 0, b_end = seg_begin(1);
    // ready words [nseg][w_words], bit i % 64 of word i / 64 = the wavefront's i-th particle: word k lives in LANE k's rm
    // (read with readlane, changed under lane == k: a few scalar instructions per push or pop, no LDS, no atomics)
    const unsigned int w_words = (w_quot + (w_rem ? 1u : 0u) + 63u) / 64u;
    unsigned long long rm = 0ull;
#pragma unroll
    for (int k = 0; k < D; ++k) hx_next[k] = 0.0;

    // ---- vector moves: VH 16-byte accesses; `ptr` points at the lane's pair 0 of the record ------
    auto st_vec = [&](auto ptr, const double (&v)[D]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < VH; ++k) {
            d2 t;
            t.x = v[2 * k];
            t.y = (2 * k + 1 < D) ? v[2 * k + 1 < D ? 2 * k + 1 : 0] : 0.0;
            ptr[k * 64] = t;
        }
    };
    auto ld_vec = [&](auto ptr, double (&v)[D]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < VH; ++k) {
            const d2 t = ptr[k * 64];
            v[2 * k] = t.x;
            if (2 * k + 1 < D) v[2 * k + 1 < D ? 2 * k + 1 : 0] = t.y;
        }
    };
    // Register-resident vectors are never updated by plain copies under a branch (the optimiser turns those into loads
    // through a selected pointer and the arrays land in scratch), and not by selects on values either where ONE
    // condition moves many values: a select of a double is two v_cndmask_b32, a move under the exec mask is one
    // v_mov_b64.  The moves are inline assembly inside a real branch, which the optimiser can neither convert back
    // into selects nor hoist; a whole group costs its moves + 3 instructions (save exec, skip-if-empty, restore).
    auto mov64 = [](double& dst, double src) __attribute__((always_inline)) {   // only ever called under a branch
        double t;
        asm volatile("v_mov_b64_e32 %0, %1" : "=v"(t) : "v"(src));   // (the join's phi makes t and dst one register)
        dst = t;
    };
    auto movv = [&](double (&dst)[D], const double (&src)[D]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < D; ++k) mov64(dst[k], src[k]);
    };
    auto movd2 = [&](d2& dst, const d2& src) __attribute__((always_inline)) {
        double a, b;
        mov64(a, src.x);
        mov64(b, src.y);
        dst.x = a; dst.y = b;
    };
    // the parked candidate of level m >= 2 / the first leaf of level l >= 3, wherever they live:
    // f(pointer) is instantiated once for LDS and once for the overflow area
    auto with_cand = [&](int m, auto&& f) __attribute__((always_inline)) {
        if (LC == 8 || m - 2 < LC) f(L + (CAND0 + (m - 2) * CREC) * 64);
        else f(O + (OCAND0 + (m - 2 - LC) * CREC) * 64);
    };
    auto with_first = [&](int l, auto&& f) __attribute__((always_inline)) {
        if (LF == 8 || l - 3 < LF) f(L + (FIRST0 + (l - 3) * FREC) * 64);
        else f(O + (OFIRST0 + (l - 3 - LF) * FREC) * 64);
    };
    auto nst_ptr = [&](int m) __attribute__((always_inline)) -> unsigned short* {
        return reinterpret_cast<unsigned short*>(L + (NST + (m >> 3)) * 64) + (m & 7);
    };
    auto is_uturn = [](double A, double B, int dir) __attribute__((always_inline)) {
        // dir > 0: minus = other, plus = current: (A < 0) || (B < 0); dir < 0: dx, roles negate
        return dir > 0 ? ((A < 0.0) || (B < 0.0)) : ((B > 0.0) || (A > 0.0));
    };

    // ---- input / output records: the contents of smcn_nuts2.hpp's, laid out PAIR-MAJOR ([transition][16-byte pair][N]:
    // the lanes of a wavefront that are at the same transition touch consecutive 16-byte chunks -- 16 cache lines per
    // vector-memory instruction where particle-major records (80 / 48 / 112 bytes apart) touched 48-64) ---------------
    gcptr2 const in2 = (gcptr2)a.in;
    gptr2 const out2 = (gptr2)a.out;
    constexpr int IPAIRS = INSZ / 2, OPAIRS = OUTSZ / 2;
    constexpr int CSZ2 = VH + 1;                       // pairs of a compact record
    // per-lane running pointers at pair 0 of the lane's current record; pair k sits k * N further, the next
    // transition's record IPAIRS * N (CSZ2 * N, OPAIRS * N) further (no 64-bit multiplies per tree)
    gcptr2 in_next = in2 + pc;
    // output: compact mode writes [b][CSZ2][N] records behind ONE [OPAIRS][N] area of full records (the last transition's)
    gptr2 out_cur = compact_mode(a) ? out2 + N * OPAIRS + p : out2 + p;
    const int64_t in_stride = N * IPAIRS, out_stride = compact_mode(a) ? N * CSZ2 : N * OPAIRS;

    // ---- per-lane state -------------------------------------------------------------------------
    int phase = live ? INIT : DONE;
    double x[D], r[D], g[D];                   // the moving edge = the current leaf
    double ex[D], er[D], eg[D];                // the edge that is not moving
    double rx[D], rr[D];                       // the accepted sample (x', r') ...
    d2 rl;                                     // ... and its (lpri, llik)
    double c0x[D], c0r[D], c1x[D], c1r[D];     // parked candidates of levels 0, 1
    d2 c0l, c1l;
    int n0 = 0, n1 = 0;
    double f1x[D], f1r[D], f2x[D], f2r[D];     // first leaves of the pending sub-trees of levels 1, 2
    double logu = 0.0, lpri0 = 0.0, llik0 = 0.0;
    const bool compact = a.logw0 != nullptr;   // transitions b < B-1 fold their forward-L weight update here (samples.py:183-196)
    double lw = compact ? ((gcptr)a.logw0)[pc] : 0.0, k0 = 0.0;
    int j = 0, i = 0, dir = 0, n = 1, nleap = 0, b = 0;
    uint32_t q = 1, qfill = 0;
    int64_t toff = 0, tlen = 0;
    bool overflow = false;
    rl.x = rl.y = c0l.x = c0l.y = c1l.x = c1l.y = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        r[k] = g[k] = ex[k] = er[k] = eg[k] = rx[k] = rr[k] = 0.0;
        c0x[k] = c0r[k] = c1x[k] = c1r[k] = f1x[k] = f1r[k] = f2x[k] = f2r[k] = 0.0;
    }

    // The next transition's record goes global -> LDS directly (global_load_lds_dwordx4: every active lane's 16
    // bytes land at base + 16 * lane, i.e. in its own pair): no registers are held across the trees in between.
    auto request = [&]() __attribute__((always_inline)) {   // the record in_next points at; then one transition further
        using lptr = __attribute__((address_space(3))) void*;
        using gvptr = const __attribute__((address_space(1))) void*;
#pragma unroll
        for (int k = 0; k <= VH; ++k)
            __builtin_amdgcn_global_load_lds((gvptr)(in_next + (VH + k) * N), (lptr)(lds3 + (PREF + k) * 64), 16, 0, 0);
        in_next += in_stride;
    };
    auto take_record = [&](bool c) __attribute__((always_inline)) {   // r, e0 of the transition about to start, from the prefetched record
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the record's LDS-DMA (a tree old) has landed
        d2 pr[VH + 1];
#ifdef SMCN_ABL_NOTAKE
#pragma unroll
        for (int k = 0; k <= VH; ++k) { pr[k].x = 0.3 + 0.1 * k; pr[k].y = -0.2; }
#else
#pragma unroll
        for (int k = 0; k <= VH; ++k) pr[k] = L[(PREF + k) * 64];
#endif
        if (c) {
#pragma unroll
            for (int k = 0; k < VH; ++k) {
                mov64(r[2 * k], pr[k].x);
                if (2 * k + 1 < D) mov64(r[2 * k + 1 < D ? 2 * k + 1 : 0], pr[k].y);
            }
            mov64(logu, pr[VH].x);        // raw; becomes H0 - e0 after the first evaluation
        }
        q = c ? 1u : q; qfill = c ? 0u : qfill; overflow = c ? false : overflow; nleap = c ? 0 : nleap;
        phase = c ? (int)INIT : phase;
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): read before the next record may overwrite the slots
    };
    auto refill = [&]() __attribute__((always_inline)) {             // draws qfill, qfill + 1 of this particle's NUTS stream
        const u32x4 o = philox4x32_10({qfill >> 1, (uint32_t)(a.particle_base + p), a.iter + (uint32_t)b, kStreamNuts},
                                      (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
        d2 t;
        t.x = u53(o.a, o.b);
        t.y = u53(o.c, o.d);
#ifdef SMCN_ABL_NOPHILOX
        t.x = 0.37 + 1e-3 * (qfill & 255u); t.y = 0.61;
#endif
        L[(RING + ((qfill >> 1) & 3u)) * 64] = t;
        qfill += 2u;
    };
    auto ring_read = [&](uint32_t qq) __attribute__((always_inline)) -> double {
        const double* ring = reinterpret_cast<const double*>(L + (RING + ((qq >> 1) & 3u)) * 64);
        return ring[qq & 1u];
    };
    // Draws fetched from the ring BEFORE the evaluation (their LDS latency hides behind it): the
    // merges of levels 0 and 1, the top-level accept and the next direction of a leaf that completes
    // its doubling.  (Scalars, not an array: a dynamically indexed array would live in scratch.)
    double up0 = 0.5, up1 = 0.5, utop = 0.5, udir = 0.5;
    bool pre_ok = false;              // utop / udir were inside the ring's fill when fetched
    auto ring_draw = [&]() __attribute__((always_inline)) -> double {   // the next draw, from the ring (or the recorded tape)
        double v;
        if constexpr (TAPE) {
            if ((int64_t)q < tlen) v = ((gcptr)a.tape)[toff + q];
            else { v = 0.5; overflow = true; }
        } else {
            if (q == qfill) refill();    // beyond what the ring held at the top of the iteration (deep merges)
            v = ring_read(q);
        }
        ++q;
        return v;
    };

    {   // x0 and the first record
        const gcptr2 rec = in_next;
#pragma unroll
        for (int k = 0; k < VH; ++k) {
            const d2 t = rec[k * N];
            x[2 * k] = t.x;
            if (2 * k + 1 < D) x[2 * k + 1 < D ? 2 * k + 1 : 0] = t.y;
        }
        request();
        if constexpr (TAPE) {
            toff = ((const __attribute__((address_space(1))) int64_t*)a.tape_off)[pc];
            tlen = ((const __attribute__((address_space(1))) int64_t*)a.tape_off)[pc + 1] - toff;
        }
        take_record(true);
        if (a.B > 1) request();      // (QUEUE: also across a segment's end -- the lane may go on with its own particle)
        if (!live) phase = DONE;
    }

    if constexpr (QUEUE) {
        // ready at the start: segment 0 of the wavefront's particles beyond the first 64
        {
            const unsigned int wi = (unsigned int)lane, lo = wi * 64u;
            const unsigned int a0 = lo < 64u ? 64u : lo, b0 = w_count < lo + 64u ? w_count : lo + 64u;
            if (wi < w_words && b0 > a0) rm = (b0 - a0 >= 64u ? ~0ull : (1ull << (b0 - a0)) - 1ull) << (a0 - lo);
        }
        n_ready = w_count - (unsigned int)kN3Block;
    }
    PROF_DECL;
#ifdef SMCN_PROFILE
    unsigned long long iters = 0;
#endif
    // the output record of the transition a lane has just ended (j doublings, accepted sample rx / rr / rl)
    auto emit_record = [&](bool last, uint32_t qdone, bool ovdone, int nldone) __attribute__((always_inline)) {
        const gptr2 orec = (compact && last) ? out2 + p : out_cur;
        out_cur += out_stride;
        d2 t;
        const unsigned long long s0 = (unsigned long long)(unsigned)nldone | ((unsigned long long)(unsigned)j << 32);
#ifndef SMCN_ABL_NOSTORE   // (ablation build: prices the record stores)
#pragma unroll
        for (int k = 0; k < VH; ++k) {   // (the record is contiguous, unlike the lane-private layouts)
            t.x = rx[2 * k];
            t.y = (2 * k + 1 < D) ? rx[2 * k + 1 < D ? 2 * k + 1 : 0] : 0.0;
            orec[k * N] = t;
        }
        PROF(12);
        if (compact && !last) {
            // COMPACT record [x', logw_b, stats]: the weight update of nuts2_post_kernel, same expressions in the
            // same order (forward L-kernel and N(0, I) momentum: L - q = -(|r'|^2 - |r0|^2) / 2 term by term)
            double k1 = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) k1 = fma(rr[k], rr[k], k1);
            const double cst = 0.5 * D * kLog2Pi;
            const double qk = -0.5 * k0 - cst;
            const double Lk = -0.5 * k1 - cst;
            const double c1 = combine_lp(rl.x, rl.y, 1.0);
            const double c0 = combine_lp(lpri0, llik0, 1.0);
            lw = lw + c1 - c0 + Lk - qk;
            t.x = lw;
            t.y = __longlong_as_double((long long)s0);
            orec[VH * N] = t;
        } else {
#pragma unroll
            for (int k = 0; k < VH; ++k) {
                t.x = rr[2 * k];
                t.y = (2 * k + 1 < D) ? rr[2 * k + 1 < D ? 2 * k + 1 : 0] : 0.0;
                orec[(VH + k) * N] = t;
            }
            orec[2 * VH * N] = rl;
            t.x = lpri0; t.y = llik0;
            orec[(2 * VH + 1) * N] = t;
            const unsigned long long s1 = (unsigned long long)qdone | ((unsigned long long)(ovdone ? 1u : 0u) << 32);
            t.x = __longlong_as_double((long long)s0);
            t.y = __longlong_as_double((long long)s1);
            orec[(2 * VH + 2) * N] = t;
        }
#endif
    };
    unsigned int it = 0u;            // (QUEUE: the iteration's number)
    for (;; ++it) {
        PROF(7);
        if constexpr (QUEUE) {
            if (__ballot(phase != DONE || have_next || own_wait) == 0ull && n_ready == 0u) break;
        }
        const bool act = phase != DONE;
#ifdef SMCN_PROFILE_TAIL   // (-DSMCN_PROFILE -DSMCN_PROFILE_TAIL=n: only the iterations with at most n trees in flight)
        PROF_ON(__popcll(__ballot(act)) <= SMCN_PROFILE_TAIL);
#endif
#ifdef SMCN_PROFILE
        if (prof_on_) ++iters;
#endif
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
            const unsigned long long am0 = __ballot(act);
            const int na0 = __popcll(am0);
#ifdef SMCN_ABL_NOTAILHELP   // (A/B build: every lane draws for itself at every population)
            if (false) {
#else
            if (Model::HAS_WIDE && (a.wide & 2) != 0 && na0 <= 16) {   // (the draws are the same bits whoever computes them)
#endif
                // Few trees left: a Philox round costs the wavefront the same for four lanes as for 64, so when a lane runs
                // low FOUR lanes draw for each tree -- lanes 4g .. 4g+3 the next four blocks of the g-th active lane's
                // stream, straight into its ring, as far as they fit -- and the next round is ~5 iterations away, not 1.
                const int avail = (int)(qfill - q);
                if (__ballot(act && avail < (need > 2 ? need : 2)) != 0ull) {
                    using u4 = unsigned int __attribute__((ext_vector_type(4)));
                    u4* const XU = reinterpret_cast<u4*>(XCH);
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(am0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am0, 0u));
                    if (act) {
                        u4 w;
                        w.x = qfill | ((unsigned)lane << 16); w.y = q;
                        w.z = (uint32_t)(a.particle_base + p); w.w = a.iter + (uint32_t)b;
                        XU[rank] = w;
                    }
                    wave_exchange_fence();
                    const int g4 = lane >> 2, h4 = lane & 3;
                    if (g4 < na0) {
                        const u4 w = XU[g4];
                        const uint32_t oq = w.x & 0xFFFFu, ol = w.x >> 16;
                        if ((int)(oq - w.y) + 2 * (h4 + 1) <= 8) {
                            const u32x4 o = philox4x32_10({(oq >> 1) + (uint32_t)h4, w.z, w.w, kStreamNuts}, (uint32_t)a.seed,
                                                          (uint32_t)(a.seed >> 32));
                            d2 t;
                            t.x = u53(o.a, o.b);
                            t.y = u53(o.c, o.d);
                            lds3[ol + (RING + (((oq >> 1) + (uint32_t)h4) & 3u)) * 64] = t;
                        }
                    }
                    wave_exchange_fence();
                    if (act) {
                        const int nfit = (8 - avail) >> 1;
                        qfill += 2u * (uint32_t)(nfit < 0 ? 0 : (nfit > 4 ? 4 : nfit));
                    }
                }
            } else {
            // ONE round per iteration whether or not a lane is short (every lane with room takes two draws): a round
            // costs the wavefront the same for one lane as for 64, and lanes that start a tree with an empty ring would
            // otherwise ask for a round of their own in each of their first iterations
            if (act && (int)(qfill - q) <= 6) refill();
            for (;;) {
                const int avail = (int)(qfill - q);
                if (__ballot(act && avail < need) == 0ull) break;
                if (act && avail <= 6) refill();
            }
            }
            // (entries beyond the ring's fill are fetched but never consumed)
            const uint32_t qt = q + (uint32_t)(phase == LEAF ? j : 0);   // INIT: the direction is the first draw
            up0 = ring_read(q); up1 = ring_read(q + 1u);
            utop = ring_read(qt); udir = ring_read(phase == LEAF ? qt + 1u : qt);
            pre_ok = (int)(qfill - qt) > 1;
        }
        PROF(0);

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
        PROF(1);
        if constexpr (Model::HAS_WIDE) {
            // few lanes left: 4 .. 64 lanes share each active lane's recurrence (the launch is as long as its longest chain;
            // 8 lanes for 5-8 stragglers: launch 2.51 -> 2.46 ms at N = 65 536)
            const unsigned long long amask = __ballot(act);
            const int nact = __popcll(amask);
            double ss = 0.0, gm = 0.0, gb = 0.0, gt = 0.0;
            if (!wide_ok || nact > 16) {     // (first: the test most iterations stop at)
                if (act) model.recur(x, ss, gm, gb, gt);
            } else if (nact > 8) model.template recur_wide<4>(x, act, amask, Yl, XCH, lane, ss, gm, gb, gt);
            else if (nact > 4) model.template recur_wide<8>(x, act, amask, Yl, XCH, lane, ss, gm, gb, gt);
            else if (nact > 2 || !wide_rows) model.template recur_wide<16>(x, act, amask, Yl, XCH, lane, ss, gm, gb, gt);
            else if (nact == 2) model.template recur_wide<32>(x, act, amask, Yl, XCH, lane, ss, gm, gb, gt);
            else model.template recur_wide<64>(x, act, amask, Yl, XCH, lane, ss, gm, gb, gt);
            if (act) model.finish(x, ss, gm, gb, gt, lpri, llik, gp, gl);
        } else {
            if (act) model.eval(x, lpri, llik, gp, gl);
        }
        PROF(2);
        double lp = lpri + phi * llik;
#pragma unroll
        for (int k = 0; k < D; ++k) g[k] = fma(phi, gl[k], gp[k]);
        if (!finite_d(lp)) {              // bridgestan.py:47-49,79-80 (a branch no lane takes, not ten selects every lane pays)
            mov64(lp, -kInf);
#pragma unroll
            for (int k = 0; k < D; ++k) mov64(g[k], -kInf);
        }

        bool start_doubling = false;
        bool tree_end = false;             // (QUEUE: the lane's tree ended in this iteration)
        const bool init = phase == INIT;      // (a lane that ends a tree below turns INIT for the NEXT iteration)
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
            d2 leafl;
            leafl.x = lpri; leafl.y = llik;
            // ---- this leaf opens the sub-trees of levels 1 .. lopen (an even leaf) ----------------
            {
                const bool opens = j > 0 && (i & 1) == 0;
                const int lopen = opens ? ((i == 0) ? j : (__ffs(i) - 1)) : 0;
                if (lopen >= 1) {
                    movv(f1x, x); movv(f1r, r);
                    if (lopen >= 2) { movv(f2x, x); movv(f2r, r); }
                }
                for (int l = 3; l <= lopen; ++l)
                    with_first(l, [&](auto fp) __attribute__((always_inline)) { st_vec(fp, x); st_vec(fp + VH * 64, r); });
            }
            // the sub-tree's candidate is kept BY REFERENCE: -1 = this leaf (x, r, lpri, llik in
            // registers), m >= 0 = the record parked at level m; it is only copied when parked one
            // level up or accepted at the top
            int csrc = -1;
            auto cand_value = [&](bool need, double (&cx)[D], double (&cr)[D], d2& cl) __attribute__((always_inline)) {   // the candidate csrc refers to
#pragma unroll
                for (int k = 0; k < D; ++k) { cx[k] = x[k]; cr[k] = r[k]; }
                cl = leafl;                          // csrc < 0: this leaf; a parked record moves in under its own mask
                if (need && csrc >= 0) {
                    if (csrc == 0) { movv(cx, c0x); movv(cr, c0r); movd2(cl, c0l); }
                    else if (csrc == 1) { movv(cx, c1x); movv(cr, c1r); movd2(cl, c1l); }
#ifndef SMCN_ABL_NOCANDLDS
                    else with_cand(csrc, [&](auto cp) __attribute__((always_inline)) {
                        double tx[D], tr[D];
                        ld_vec(cp, tx); ld_vec(cp + VH * 64, tr);
                        const d2 tl = cp[2 * VH * 64];
                        movv(cx, tx); movv(cr, tr); movd2(cl, tl);
                    });
#endif
                }
            };
            // one merge (nuts.py:136-148) of the parked first half (count nfirst) whose sub-tree began at
            // the leaf (fx, fr), drawing u
            auto merge = [&](int m, double u, int nfirst, const double (&fx)[D], const double (&fr)[D]) __attribute__((always_inline)) {
                const int den = (nfirst + nsub) > 1 ? (nfirst + nsub) : 1;
                const bool keep = !(fma(u, (double)den, -(double)nsub) < 0.0);   // keep the first half's candidate
                csrc = keep ? m : csrc;
                nsub += nfirst;              // :146
                double A = 0.0, B = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double d = x[k] - fx[k];
                    A = fma(d, fr[k], A);
                    B = fma(d, r[k], B);
                }
                ssub = is_uturn(A, B, dir);  // :148
            };
            PROF(3);
            // ---- merges (nuts.py:134-148), the top level (:99-105) being level j ------------------
            // The level loop only MERGES (second halves); how it ends says what happens to the result:
            // a stopped sub-tree unwinds (1), level j is the top (2), a first half is parked at level m (3).
            // The candidate is materialised once, after the loop, for the park or the top-level accept.
            int m = 0, how = 0;
            auto ends = [&]() __attribute__((always_inline)) -> int {   // how the loop ends at level m, 0 = merge and go on
                return ssub ? 1 : (m == j ? 2 : ((((i >> m) & 1) == 0) ? 3 : 0));
            };
            how = ends();
            if (how == 0) {          // level 0 (registers)
                double u;
                if constexpr (TAPE) u = ring_draw(); else { u = up0; ++q; }
                merge(0, u, n0, f1x, f1r);
                m = 1;
                how = ends();
            }
            if (how == 0) {          // level 1 (registers)
                double u;
                if constexpr (TAPE) u = ring_draw(); else { u = up1; ++q; }
                merge(1, u, n1, f2x, f2r);
                m = 2;
                how = ends();
            }
            while (how == 0) {       // levels >= 2 (LDS, then the overflow area)
                double fx[D], fr[D];
#ifdef SMCN_ABL_NOMERGELDS   // (ablation builds price one piece of the bookkeeping each: wrong trees, per-iteration cycles only)
                const int nfirst = 1;
#pragma unroll
                for (int k = 0; k < D; ++k) { fx[k] = ex[k]; fr[k] = er[k]; }
                const double u = 0.5; ++q;
#else
                const int nfirst = (int)*nst_ptr(m);
                with_first(m + 1, [&](auto fp) __attribute__((always_inline)) { ld_vec(fp, fx); ld_vec(fp + VH * 64, fr); });
                const double u = ring_draw();     // :142, always
#endif
                merge(m, u, nfirst, fx, fr);
                ++m;
                how = ends();
            }
            PROF(8);
            bool done = false, stop = false, acc = false;
            if (how == 1) {
                // unwinding: every ancestor whose SECOND half stopped still draws (:142)
                q += (uint32_t)__popc((unsigned)(i >> m) & ((1u << (j - m)) - 1u));
                done = true; stop = true;
            } else if (how == 2) {
                // top level: accept with prob min(1, n'/n) (:99), U-turn on the outer edges (:105)
                double u;
                if (TAPE || !pre_ok) u = ring_draw(); else { u = utop; ++q; }
                acc = nsub >= n || fma(u, (double)n, -(double)nsub) < 0.0;
                double A = 0.0, B = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double d = x[k] - ex[k];
                    A = fma(d, er[k], A);
                    B = fma(d, r[k], B);
                }
                stop = is_uturn(A, B, dir);
                done = true;
            }
            PROF(9);
            {   // the candidate goes to the parked slot of level m (3) or becomes the accepted sample (2, accepted)
                double cx[D], cr[D];
                d2 cl;
                const bool park = how == 3;
                cand_value(park || acc, cx, cr, cl);
                if (acc) { movv(rx, cx); movv(rr, cr); movd2(rl, cl); }
                if (park) {
                    if (m == 0) { movv(c0x, cx); movv(c0r, cr); movd2(c0l, cl); }
                    else if (m == 1) { movv(c1x, cx); movv(c1r, cr); movd2(c1l, cl); }
                    else {
                        with_cand(m, [&](auto crec) __attribute__((always_inline)) { st_vec(crec, cx); st_vec(crec + VH * 64, cr); crec[2 * VH * 64] = cl; });
                        *nst_ptr(m) = (unsigned short)nsub;
                    }
                }
                n0 = (park && m == 0) ? nsub : n0;
                n1 = (park && m == 1) ? nsub : n1;
            }
            PROF(4);
            if (!done) {
                ++i;
            } else {
                n += nsub;                   // :103  (unused after a stop)
                ++j;
                if (stop || j > a.max_depth) {   // :89,109 -> emit the output record, start the next transition
                    if constexpr (QUEUE) {
                        tree_end = true;         // (handled below, where every lane of the wavefront is, with the hand-overs)
                    } else {
                        const bool more = b + 1 < a.B;
                        // the next transition's record first: its loads are a whole tree old, so this wait
                        // does not land on the stores below
                        const int bdone = b;
                        const uint32_t qdone = q;
                        const bool ovdone = overflow;
                        const int nldone = nleap;
                        if (more) movv(x, rx);        // continue from the sample just drawn
                        phase = DONE;
                        take_record(more);
                        b = more ? bdone + 1 : b;
                        PROF(10);
                        emit_record(!more, qdone, ovdone, nldone);
                        PROF(13);
#ifndef SMCN_ABL_NOLOAD
                        if (more && bdone + 2 < a.B) request();
#endif
                    }
                } else {
                    start_doubling = true;
                }
            }
        }
        if constexpr (QUEUE) {
            // ---- tree ends and the wavefront's ready jobs, for all lanes at once.  Everything this block consumes -- the
            // next transition's record, a job's start state -- was asked for by this block an iteration (or a tree) ago, and
            // everything it issues (record stores, hand-over stores, loads) comes after its one wait: vmcnt counts loads
            // and stores in order, so a wait anywhere else would land on this block's young stores.
            if (__ballot(tree_end || (phase == DONE && (have_next || own_wait || n_ready != 0u))) != 0ull) {
                __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
                const bool idle = phase == DONE;                    // (no tree in this iteration)
                const bool sd = b + 1 >= b_end;                     // (at a tree's end: it was the segment's last transition)
                const bool last = tree_end && b + 1 >= a.B;        // ... the block's
                // a lane starts a tree only in an iteration that keeps its trees in step with the wavefront's (always so after
                // a tree that ran its 2^depth iterations; after one that stopped early the lane sits out up to 3 iterations)
                const bool in_step = ((it + 1u) & (unsigned int)(a.seg_align - 1)) == 0u;
                const bool to_next = have_next && ((tree_end && sd) || idle);       // goes over to its next job ...
                const bool take = to_next && in_step;
                const bool push = tree_end && have_next && sd && !last;            // its own particle is left to the wavefront
                const bool to_own = (tree_end && !(have_next && sd) && !last) || (idle && own_wait);   // ... or on with its own particle
                const bool cont = to_own && in_step;
                own_wait = to_own && !in_step;
                const bool seg_done = cont && sd;
                const int bdone = b;
                const uint32_t qdone = q;
                const bool ovdone = overflow;
                const int nldone = nleap;
                if (cont) movv(x, rx);            // continue from the sample just drawn
                phase = tree_end ? (int)DONE : phase;
                take_record(cont || take);
                PROF(10);
                if (tree_end) emit_record(last, qdone, ovdone, nldone);
                const unsigned long long pm = __ballot(push);
                if (pm != 0ull) {
                    if (push) {
                        // (x', running log-weight) for whichever lane takes the particle's next segment -- a record of its own
                        // per (particle, segment), written once and read once per launch --, and that segment's bit
                        const gptr2 ho = (gptr2)a.handover + ((size_t)p * (nseg - 1) + (size_t)seg) * (VH + 1);
                        d2 t;
#pragma unroll
                        for (int k = 0; k < VH; ++k) {
                            t.x = rx[2 * k];
                            t.y = (2 * k + 1 < D) ? rx[2 * k + 1 < D ? 2 * k + 1 : 0] : 0.0;
                            ho[k] = t;
                        }
                        t.x = lw; t.y = 0.0;
                        ho[VH] = t;
                    }
                    const unsigned int ip = (unsigned int)(p - w_start);
                    const unsigned int pw = (unsigned int)(seg + 1) * w_words + (ip >> 6), pb = ip & 63u;
                    for (unsigned long long m = pm; m != 0ull; m &= m - 1ull) {      // (one turn per lane that hands on)
                        const int l = (int)__builtin_ctzll(m);
                        const unsigned int wu = (unsigned int)__builtin_amdgcn_readlane((int)pw, l);
                        const unsigned int bu = (unsigned int)__builtin_amdgcn_readlane((int)pb, l);
                        if ((unsigned int)lane == wu) rm |= 1ull << bu;
                    }
                    n_ready += (unsigned int)__popcll(pm);
                }
                if (take) {
#pragma unroll
                    for (int k = 0; k < D; ++k) mov64(x[k], hx_next[k]);
                    mov64(lw, lw_next);
                    if constexpr (TAPE) { toff = toff_next; tlen = tlen_next; }
                    seg = seg_next;
                    p = w_start + (int64_t)i_next;
                    out_cur = (compact_mode(a) ? out2 + N * OPAIRS + p : out2 + p) + (int64_t)seg_begin(seg) * out_stride;
                    have_next = false;
                }
                // (cont across a segment's end: nothing was ready when the lane looked, its own next segment is its job)
                seg = seg_done ? seg + 1 : seg;
                b = take ? seg_begin(seg) : (cont ? bdone + 1 : b);
                b_end = (take || seg_done) ? seg_begin(seg + 1) : b_end;
                const bool runs = cont || take;
                PROF(13);
                // the lanes that start their segment's last transition, and the lanes without a job, take the ready jobs, the
                // least advanced particles first: the r-th such lane the r-th set bit of the ready words
                const bool want = !have_next && ((runs && b + 1 >= b_end) || (!runs && phase == DONE && !own_wait));
                const unsigned long long wm = __ballot(want);
                bool got = false;
                if (wm != 0ull && n_ready != 0u) {
                    unsigned int job = 0u;
                    for (unsigned long long m = wm; m != 0ull && n_ready != 0u; m &= m - 1ull) {   // (one turn per lane that takes a job)
                        const int l = (int)__builtin_ctzll(m);
                        const int wu = (int)__builtin_ctzll(__ballot(rm != 0ull));                 // the lowest word that has a bit
                        const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)rm, wu);
                        const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(rm >> 32), wu);
                        const unsigned int bu = lo ? (unsigned int)__builtin_ctz(lo) : 32u + (unsigned int)__builtin_ctz(hi);
                        if (lane == wu) rm &= rm - 1ull;                                             // ... loses its lowest one
                        if (lane == l) { job = (unsigned int)wu * 64u + bu; got = true; }
                        --n_ready;
                    }
                    if (got) {
                        const unsigned int wi = job >> 6, sg = wi / w_words;
                        seg_next = (int)sg;
                        i_next = (wi - sg * w_words) * 64u + (job & 63u);
                        const int64_t pn = w_start + (int64_t)i_next;
                        in_next = in2 + pn + (int64_t)seg_begin(seg_next) * in_stride;
                        // its start state: x0 = pairs 0 .. VH-1 of its first record and the start weight (segment 0), or the
                        // hand-over record (x', running log-weight) of the segment before
                        const gcptr2 src = seg_next == 0 ? in_next
                                                         : (gcptr2)a.handover + ((size_t)pn * (nseg - 1) + (size_t)(seg_next - 1)) * (VH + 1);
                        const int64_t sstep = seg_next == 0 ? N : 1;
#pragma unroll
                        for (int k = 0; k < VH; ++k) {
                            const d2 t = src[k * sstep];
                            hx_next[2 * k] = t.x;
                            if (2 * k + 1 < D) hx_next[2 * k + 1 < D ? 2 * k + 1 : 0] = t.y;
                        }
                        if (seg_next == 0) lw_next = compact ? ((gcptr)a.logw0)[pn] : 0.0;
                        else lw_next = src[VH].x;
                        if constexpr (TAPE) {
                            toff_next = ((const __attribute__((address_space(1))) int64_t*)a.tape_off)[pn];
                            tlen_next = ((const __attribute__((address_space(1))) int64_t*)a.tape_off)[pn + 1] - toff_next;
                        }
                        have_next = true;
                    }
                }
#ifndef SMCN_ABL_NOLOAD
                // the prefetch slots get the first record of the job just taken, or the lane's own next record
                if (got || (runs && b + 1 < a.B)) request();
#endif
            }
        }
        PROF(11);
        {
            // ---- nuts.py:66-87 ----------------------------------------------------------------
            double kin = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) kin = fma(r[k], r[k], kin);
#ifdef SMCN_ABL_INITMOVS   // (ablation: what do moves under a FEW-lane exec mask cost?  the init block's moves repeated)
            if (init) {
#pragma unroll
                for (int rep = 0; rep < SMCN_ABL_INITMOVS; ++rep) { movv(rx, x); movv(rr, r); mov64(k0, kin); mov64(lpri0, lpri); }
            }
#endif
            if (init) {
                mov64(logu, (lp - 0.5 * kin) - logu);          // H0 - Exp(1)
                mov64(k0, kin);                                // |r0|^2: q = N(r0; 0, I) of the weight update
                mov64(lpri0, lpri); mov64(llik0, llik);        // the record's start density
                movv(rx, x); movv(rr, r);
                d2 il;
                il.x = lpri; il.y = llik;
                movd2(rl, il);
            }
            j = init ? 0 : j; n = init ? 1 : n;
            dir = init ? 0 : dir;                 // both edges are (x0, r0, g0)
            start_doubling = start_doubling || init;
            phase = init ? (int)LEAF : phase;
        }
        PROF(5);
        {
            // ---- nuts.py:91: direction; the moving state becomes that edge --------------------------
            int nd = dir;
            if (start_doubling) {
                double u;
                if (TAPE || !pre_ok) u = ring_draw(); else { u = udir; ++q; }
                nd = (u < 0.5) ? 1 : -1;
            }
            const bool first = start_doubling && dir == 0;              // both edges are the start state
            const bool swap = start_doubling && dir != 0 && nd != dir;  // the moving edge and the parked one trade places
            if (first || swap) {            // the parked edge takes the moving state; on a swap the moving state takes the parked edge
                double tx[D], tr[D], tg[D];
#pragma unroll
                for (int k = 0; k < D; ++k) { tx[k] = ex[k]; tr[k] = er[k]; tg[k] = eg[k]; }
                movv(ex, x); movv(er, r); movv(eg, g);
                if (swap) { movv(x, tx); movv(r, tr); movv(g, tg); }
            }
            dir = nd;
            i = start_doubling ? 0 : i;
        }
        PROF(6);
        if constexpr (!QUEUE) {
            if (__ballot(phase != DONE) == 0ull) break;
        }
    }
    PROF_FLUSH(a);
#ifdef SMCN_PROFILE
    if (lane == 0) {   // prof[14]: wave-iterations summed, prof[15]: the longest wave
        atomicAdd(&a.prof[14], iters);
        atomicMax(&a.prof[15], iters);
    }
#endif
}

}  // namespace smcn
