// A/B builds only (-DSMCN_VARIANTS): group functors the product no longer uses -- arma on lane groups (scan of the
// carries; the product evaluates arma with one lane per particle, smcn_nuts3.hpp) and PRMwCD with the state replicated
// on every lane of the group (the product distributes it: PrmwcdDistModel).
#pragma once
#include "smcn_models.hpp"

namespace smcn {

// ---------------------------------------------------------------------------
// ARMA(1,1), x = (mu, beta, theta, s), sigma = exp(s).  mdata = [T, y...].
//
// The T-step recurrence err_t = c_t - theta * err_{t-1} is first-order linear,
// so the G lanes each own S consecutive time steps and the carries are joined
// with a log2(G)-stage scan (all lanes share the multiplier (-theta)^S).  The
// gradient is the adjoint recurrence a_t = err_t - theta * a_{t+1}, run the same
// way in the opposite direction:
//   d/dmu    sum err^2/2 = -sum_t a_t * [1  (t>1) | 1+beta (t=1)]
//   d/dbeta  sum err^2/2 = -sum_t a_t * [y_{t-1}   | mu     (t=1)]
//   d/dtheta sum err^2/2 = -sum_t a_t * err_{t-1}
// GE*S >= T; EXACT means GE*S == T (no padding predicates are generated).
// Padding, if any, sits in front of t=1 where the recurrence is identically 0.
// ---------------------------------------------------------------------------
template <int GE_, int S_, bool EXACT, int PAIR = 1>
struct ArmaModel {
    // GE lanes evaluate one particle's recurrence together; the particle's STATE is replicated on
    // G = GE / PAIR lanes only, so a wavefront carries 64 / G particles: the PAIR particles of an
    // evaluation group take turns on the GE lanes for the recurrence (the only part that needs
    // them) while everything per-particle-scalar -- the density tail here, the whole tree
    // bookkeeping in the kernel -- is issued once for 64 / G particles instead of 64 / GE.
    static constexpr int GE = GE_, G = GE_ / PAIR, DL = 4, S = S_, SHARED = GE_ * S_ + 2, MIN_WAVES = 2;
    static constexpr int LDS_LEVELS = 2;                       // v1 kernel (hybrid stack), unused for arma
    static constexpr int N2_LDS_LEVELS = (GE_ / PAIR >= 8) ? 10 : 4;   // v2 kernel: tree-stack levels in LDS
    static_assert(PAIR == 1 || PAIR == 2, "one or two particles per evaluation group");
    static constexpr bool DIST = false;
    int T, pad, lg;   // lg: lane within the EVALUATION group
    const double* y;  // block-shared LDS: y[k] = y_{t0+k-1} (0 outside 1..T), t0 = first step of this lane

    __device__ int dim() const { return 4; }
    __device__ void init(const double* md, int, double* shared) {
        lg = (int)(threadIdx.x & (GE - 1));
        T = (int)md[0];
        pad = GE * S - T;
        for (int idx = threadIdx.x; idx <= GE * S; idx += blockDim.x) {
            const int t = idx - 1 - pad;         // 0-based time index held at shared[idx]
            shared[idx] = (t >= 0 && t < T) ? md[1 + t] : 0.0;
        }
        y = shared + lg * S;
        __syncthreads();
    }

    template <int P>
    static __device__ __forceinline__ double ipow(double a) {
        if constexpr (P == 0) return 1.0;
        else if constexpr (P == 1) return a;
        else {
            const double h = ipow<P / 2>(a);
            if constexpr (P % 2) return h * h * a;
            else return h * h;
        }
    }

    // sums over t of err^2 and of the three adjoint products, for the parameters on THIS lane's
    // evaluation group (all GE lanes pass the same mu, beta, theta)
    __device__ __forceinline__ void recurrence(double mu, double beta, double theta, double& ss, double& gm,
                                               double& gb, double& gt) const {
        constexpr int G = GE;
        const double nth = -theta;
        const int kfirst = pad - lg * S;  // step index (in this lane) of t = 1, if in [0, S)

        // ---- forward, pass 1: c_k and the lane-local recurrence from a zero carry
        double c[S];
        double e = 0.0;
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const bool first = EXACT ? (k == 0 && lg == 0) : (k == kfirst);
            const double yp = first ? mu : y[k];          // arma.stan:25 nu[1] = mu + beta*mu
            double ck = (y[k + 1] - mu) - beta * yp;
            if constexpr (!EXACT) ck = (k >= kfirst) ? ck : 0.0;
            c[k] = ck;
            e = fma(nth, e, ck);
        }
        // ---- scan of lane carries: incl_l = sum_{j<=l} A^(l-j) B_j, A = (-theta)^S
        const double A = ipow<S>(nth);
        double incl = e, Ak = A;
        if constexpr (G >= 2)  { incl = fma(Ak, group_shift_up<G, 1>(incl, lg), incl);  Ak *= Ak; }
        if constexpr (G >= 4)  { incl = fma(Ak, group_shift_up<G, 2>(incl, lg), incl);  Ak *= Ak; }
        if constexpr (G >= 8)  { incl = fma(Ak, group_shift_up<G, 4>(incl, lg), incl);  Ak *= Ak; }
        if constexpr (G >= 16) { incl = fma(Ak, group_shift_up<G, 8>(incl, lg), incl);  Ak *= Ak; }
        if constexpr (G >= 32) { incl = fma(Ak, group_shift_up<G, 16>(incl, lg), incl); Ak *= Ak; }
        if constexpr (G >= 64) { incl = fma(Ak, group_shift_up<G, 32>(incl, lg), incl); }
        double carry = 0.0;
        if constexpr (G >= 2) carry = group_shift_up<G, 1>(incl, lg);  // err just before this lane's block
        // ---- forward, pass 2: true err_k (overwrites c[k]); sum of squares
        e = carry;
        ss = 0.0;
#pragma unroll
        for (int k = 0; k < S; ++k) {
            e = fma(nth, e, c[k]);
            c[k] = e;
            ss = fma(e, e, ss);
        }
        // ---- backward, pass 1: lane-local adjoint from a zero carry
        double a = 0.0;
#pragma unroll
        for (int k = S - 1; k >= 0; --k) a = fma(nth, a, c[k]);
        incl = a; Ak = A;
        if constexpr (G >= 2)  { incl = fma(Ak, group_shift_down<G, 1>(incl, lg), incl);  Ak *= Ak; }
        if constexpr (G >= 4)  { incl = fma(Ak, group_shift_down<G, 2>(incl, lg), incl);  Ak *= Ak; }
        if constexpr (G >= 8)  { incl = fma(Ak, group_shift_down<G, 4>(incl, lg), incl);  Ak *= Ak; }
        if constexpr (G >= 16) { incl = fma(Ak, group_shift_down<G, 8>(incl, lg), incl);  Ak *= Ak; }
        if constexpr (G >= 32) { incl = fma(Ak, group_shift_down<G, 16>(incl, lg), incl); Ak *= Ak; }
        if constexpr (G >= 64) { incl = fma(Ak, group_shift_down<G, 32>(incl, lg), incl); }
        double carryb = 0.0;
        if constexpr (G >= 2) carryb = group_shift_down<G, 1>(incl, lg);
        // ---- backward, pass 2: true adjoints and the three sums
        a = carryb;
        gm = 0.0; gb = 0.0; gt = 0.0;
#pragma unroll
        for (int k = S - 1; k >= 0; --k) {
            a = fma(nth, a, c[k]);
            const bool first = EXACT ? (k == 0 && lg == 0) : (k == kfirst);
            const double yp = first ? mu : y[k];
            const double ep = (k > 0) ? c[k - 1] : carry;   // err_{t-1}; 0 in front of t=1
            double am = a;
            if constexpr (!EXACT) am = (k >= kfirst) ? a : 0.0;
            gm += first ? am * (1.0 + beta) : am;
            gb = fma(am, yp, gb);
            gt = fma(am, ep, gt);
        }
        ss = group_sum<G>(ss);
        gm = group_sum<G>(gm);
        gb = group_sum<G>(gb);
        gt = group_sum<G>(gt);
    }

    __device__ void eval(const double (&x)[4], double& lpri, double& llik, double (&gp)[4],
                         double (&gl)[4]) const {
        const double mu = x[0], beta = x[1], theta = x[2], s = x[3];
        double ss, gm, gb, gt;
        if constexpr (PAIR == 1) {
            recurrence(mu, beta, theta, ss, gm, gb, gt);
        } else {
            // lanes [0, G) of the evaluation group hold particle A, lanes [G, 2G) particle B
            const bool hi = lg >= G;
            double s0, m0, b0, t0, s1, m1, b1, t1;
            recurrence(group_read<GE>(mu, 0), group_read<GE>(beta, 0), group_read<GE>(theta, 0), s0, m0, b0, t0);
            recurrence(group_read<GE>(mu, G), group_read<GE>(beta, G), group_read<GE>(theta, G), s1, m1, b1, t1);
            ss = hi ? s1 : s0; gm = hi ? m1 : m0; gb = hi ? b1 : b0; gt = hi ? t1 : t0;
        }

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
        llik = -0.5 * T * kLog2Pi - T * s - 0.5 * ss * w;
        gl[0] = w * gm;
        gl[1] = w * gb;
        gl[2] = w * gt;
        gl[3] = ss * w - (double)T;
    }
};

// ---------------------------------------------------------------------------
// PRMwCD: Poisson regression on a Gaussian-kernel design with an
// exponential-power prior (stan_models/PRMwCD/PRMwCD.stan:11-38).
// x = (Beta_1..Beta_M, g), Gamma = exp(g); mdata = [Nobs, M, C, q, y.., X..],
// M = C + 1.  The design matrix (rows padded to RS doubles) and y sit in
// block-shared LDS; lane lg owns observations lg, lg+G, .. and the prior terms
// of Beta_j with (j-1) % G == lg, and a butterfly sums the 1 + D partials.
// ---------------------------------------------------------------------------
template <int G_, int NOBS, int C_>
struct PrmwcdModel {
    static constexpr int G = G_, C = C_, M = C_ + 1, DL = C_ + 2, RS = (C_ + 1 + 1) & ~1;
    static constexpr int SHARED = NOBS * RS + 2 * NOBS, MIN_WAVES = 1, LDS_LEVELS = 2, N2_LDS_LEVELS = 10;
    static constexpr bool DIST = false;
    static constexpr int S = (NOBS + G - 1) / G;
    int lg;
    double q;
    const double* X;  // [NOBS][RS] in LDS
    const double* y;  // [NOBS]     in LDS

    __device__ int dim() const { return DL; }
    __device__ void init(const double* md, int lg_, double* shared) {
        lg = lg_;
        q = md[3];
        for (int t = threadIdx.x; t < NOBS * RS; t += blockDim.x) {
            const int i = t / RS, j = t - i * RS;
            shared[t] = (j < C) ? md[4 + NOBS + i * C + j] : 0.0;
        }
        for (int t = threadIdx.x; t < NOBS; t += blockDim.x) {
            shared[NOBS * RS + t] = md[4 + t];
            shared[NOBS * RS + NOBS + t] = lgamma(md[4 + t] + 1.0);   // data-only term of poisson_lpmf
        }
        X = shared;
        y = shared + NOBS * RS;
        __syncthreads();
    }

    __device__ void eval(const double (&x)[DL], double& lpri, double& llik, double (&gp)[DL],
                         double (&gl)[DL]) const {
        const double g = x[M];
        const double eg = exp(-g);
        // ---- likelihood partials of this lane's observations (PRMwCD.stan:24-33)
        double ll = 0.0;
#pragma unroll
        for (int j = 0; j < DL; ++j) gl[j] = 0.0;
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const int i = lg + G * k;
            const bool live = i < NOBS;
            const double* row = X + (live ? i : 0) * RS;
            double eta = x[0];
#pragma unroll
            for (int j = 0; j < C; ++j) eta = fma(x[j + 1], row[j], eta);
            const double mu = exp(eta);
            const double yi = live ? y[i] : 0.0;
            double term;
            if (__builtin_isinf(mu)) term = -kInf;                       // poisson_lpmf(y | inf)
            else if (mu == 0.0 && yi != 0.0) term = -kInf;               // lambda == 0, n != 0
            else term = (yi == 0.0 ? 0.0 : yi * eta) - mu - (live ? y[NOBS + i] : 0.0);
            const double d = live ? (yi - mu) : 0.0;
            ll += live ? term : 0.0;
            gl[0] += d;
#pragma unroll
            for (int j = 0; j < C; ++j) gl[j + 1] = fma(d, row[j], gl[j + 1]);
        }
        // ---- prior partials: inv_gamma(Gamma | 2, 1.3) + Jacobian on lane 0;
        //      exponential-power terms of Beta_2..Beta_M spread over the lanes (:36-38)
        double lp = (lg == 0) ? (2.0 * 0.26236426446749105203 - 3.0 * g - 1.3 * eg + g) : 0.0;  // lgamma(2) = 0
        double dg = (lg == 0) ? (-3.0 + 1.3 * eg + 1.0) : 0.0;
        const double egq = (q == 0.5) ? sqrt(eg) : pow(eg, q);
#pragma unroll
        for (int j = 1; j < M; ++j) {
            const bool mine = ((j - 1) % G) == lg;
            const double ab = fabs(x[j]);
            const double apow = (q == 0.5) ? sqrt(ab) : pow(ab, q);   // |Beta_j|^q
            const double p = apow * egq;                               // (|Beta_j| / Gamma)^q
            lp += mine ? (-g - p) : 0.0;
            dg += mine ? (-1.0 + q * p) : 0.0;
            const double sgn = (x[j] > 0.0) ? 1.0 : ((x[j] < 0.0) ? -1.0 : 0.0);
            gp[j] = mine ? (-q * sgn * (apow / ab) * egq) : 0.0;      // -q sgn |b|^(q-1) e^(-gq)
        }
        gp[0] = 0.0;
        gp[M] = dg;
        llik = group_sum<G>(ll);
        lpri = group_sum<G>(lp);
#pragma unroll
        for (int j = 0; j < DL; ++j) {
            gl[j] = group_sum<G>(gl[j]);
            if (j >= 1) gp[j] = group_sum<G>(gp[j]);
        }
    }
};

// ---------------------------------------------------------------------------
// PRMwCD with ONE LANE PER PARTICLE (round 5, A/B: SMCN_PRMWCD_LANE): all 13 coordinates on the lane, the design matrix read
// by SCALAR loads (a row is wave-uniform: it enters the FMAs as an SGPR operand, as the arma series does), no cross-lane
// traffic at all.  Per evaluation 100 x ~50 vector instructions for 64 particles -- 78 per particle-leapfrog where the
// 8-lane functor issues ~200 -- at the price of 64 trees in lock step.  mdata = [N, M, Clength, q, y_1..y_N, X (N x C)].
// ---------------------------------------------------------------------------
template <int NOBS, int C_, int LEVELS = 1, bool LK = false>
struct PrmwcdLaneModel {
    static constexpr bool LANE_KERNEL = LK;                 // nuts_lane_kernel (smcn_nuts_lane.hpp) instead of nuts_kernel with G = 1
    static constexpr int G = 1, C = C_, M = C_ + 1, D_ = C_ + 2, DL = C_ + 2;
    static constexpr int RS = (C_ + 2) & ~1;                // table row: X_i1 .. X_iC, (pad,) y_i
    // (LK: the row table lives in LDS -- a scalar load of a row takes ~400 cycles here, twice an observation's arithmetic, and
    //  102 scalar registers do not hold a prefetch distance of two rows; an LDS broadcast read returns in ~100)
    static constexpr int SHARED = LK ? (NOBS + 2) * RS : 0, MIN_WAVES = 1, LDS_LEVELS = LEVELS;
    static constexpr bool DIST = true;
    static constexpr bool HYBRID_ALWAYS = true;
    static constexpr bool TWO_PHASE = true;               // nuts_kernel parks its long trees for nuts_fin_kernel (from the slot)
    static constexpr int STEP_ALIGN = 16;
    using cptr = const __attribute__((address_space(4))) double*;
    int nobs;
    cptr tab;     // [nobs + 2][RS], 128-byte aligned, behind the caller's data (smcn_ctx_create)
    const double* tabl;   // LK: the same table in LDS
    double lgsum; // sum_i lgamma(y_i + 1): the data-only term of the Poisson log-likelihood

    __device__ int dim() const { return D_; }
    __device__ void init(const double* md, int, double* shared) {
        const cptr m = (cptr)md;
        nobs = __builtin_amdgcn_readfirstlane((int)m[0]);
        const int len = 4 + nobs * (C + 1);
        tab = m + (len + 15) / 16 * 16;
        tabl = nullptr;
        if constexpr (LK) {
            const int cnt = (nobs + 2) * RS < SHARED ? (nobs + 2) * RS : SHARED;
            for (int t = threadIdx.x; t < cnt; t += blockDim.x) shared[t] = tab[t];
            tabl = shared;
            __syncthreads();
        }
        double s = 0.0;
        for (int i = 0; i < nobs; ++i) s += lgamma(m[4 + i] + 1.0);
        lgsum = s;
    }
    __device__ void eval(const double (&x)[DL], double& lpri, double& llik, double (&gp)[DL], double (&gl)[DL]) const {
        const double g = x[M];
        const double egq = exp_fast(-0.5 * g), eg = egq * egq;
        double acc[M], ll = 0.0, mumax = 0.0, mmin = 1.0;
        [[maybe_unused]] int kmin = 0, kmax = 0;
#pragma unroll
        for (int j = 0; j < M; ++j) acc[j] = 0.0;
        // one observation: its row R is wave-uniform (scalar registers), everything else is this lane's particle
        auto head = [&](const double (&R)[RS]) __attribute__((always_inline)) -> double {
            return fma(x[1], R[0], x[0]);
        };
        auto rest = [&](const double (&R)[RS], double e) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 1; j < C; ++j) e = fma(x[j + 1], R[j], e);
#ifdef SMCN_LANE_FASTEXP
            // e^t without v_rndne / v_cvt / v_ldexp: k from the magic-number add (its low dword IS the integer), 2^k by an integer
            // add into the exponent field -- valid while the result is a normal number; kmin / kmax flag everything else
            const double tm = fma(e, 1.4426950408889634074, 6755399441055744.0);       // 1.5 * 2^52
            const double kd = tm - 6755399441055744.0;
            const int ki = __double2loint(tm);
            double rr = fma(-kd, 6.93147180369123816490e-01, e);
            rr = fma(-kd, 1.90821492927058770002e-10, rr);
            double pp;
            asm("v_fma_f64 %0, %1, %2, %3\n\t"
                "v_fma_f64 %0, %0, %2, %4\n\t"
                "v_fma_f64 %0, %0, %2, %5\n\t"
                "v_fma_f64 %0, %0, %2, %6\n\t"
                "v_fma_f64 %0, %0, %2, %7\n\t"
                "v_fma_f64 %0, %0, %2, %8\n\t"
                "v_fma_f64 %0, %0, %2, %9\n\t"
                "v_fma_f64 %0, %0, %2, %10\n\t"
                "v_fma_f64 %0, %0, %2, %11"
                : "=&v"(pp)
                : "v"(2.08767569878680989792e-09), "v"(rr), "s"(2.50521083854417187751e-08), "s"(2.75573192239858906526e-07),
                  "s"(2.75573192239858906526e-06), "s"(2.48015873015873015873e-05), "s"(1.98412698412698412698e-04),
                  "s"(1.38888888888888888889e-03), "s"(8.33333333333333333333e-03), "s"(4.16666666666666666667e-02),
                  "s"(1.66666666666666666667e-01));
            pp = fma(pp, rr, 0.5);
            pp = fma(pp, rr, 1.0);
            pp = fma(pp, rr, 1.0);
            const double mu = __hiloint2double(__double2hiint(pp) + (ki << 20), __double2loint(pp));
            kmin = ki < kmin ? ki : kmin;
            kmax = ki > kmax ? ki : kmax;
#else
            const double mu = exp_fast_s(e);
#endif
            const double yi = R[RS - 1];
            double t1, term;
            {
#pragma clang fp contract(off)
                t1 = yi * e;                                   // (0 for y = 0, as the reference's select)
                term = t1 - mu;
            }
            const double d = yi - mu;
#ifndef SMCN_LANE_NOEDGE
            mumax = fmax(mumax, mu);
            mmin = fmin(mmin, mu + (yi == 0.0 ? 1.0 : 0.0));
#endif
            ll += term;
            acc[0] += d;
#pragma unroll
            for (int j = 0; j < C; ++j) acc[j + 1] = fma(d, R[j], acc[j + 1]);
        };
        if constexpr (LK) {
            // rows by LDS broadcast reads (every lane the same address), one observation ahead; LDS returns in order, so the
            // compiler's waits are exact
            using d2 = double __attribute__((ext_vector_type(2)));
            using lds2 = const __attribute__((address_space(3))) d2*;
            const lds2 T = (lds2)tabl;
            double A[RS], B[RS];
            auto ldrow = [&](double (&R)[RS], int row) __attribute__((always_inline)) {
#pragma unroll
                for (int k2 = 0; k2 < RS / 2; ++k2) { const d2 t = T[row * (RS / 2) + k2]; R[2 * k2] = t.x; R[2 * k2 + 1] = t.y; }
            };
            ldrow(A, 0);
            int i = 0;
            for (; i + 1 < nobs; i += 2) {
                ldrow(B, i + 1);
                rest(A, head(A));
                __builtin_amdgcn_sched_barrier(0);
                ldrow(A, i + 2);
                rest(B, head(B));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (i < nobs) rest(A, head(A));
        } else {
        // Two register sets take turns; a row is asked for one observation ahead of its use, right behind the wait for the
        // previous one (scalar loads return out of order, so a wait is for all of them: smcn_nuts3.hpp's recurrence)
        double A[RS], B[RS];
#pragma unroll
        for (int k = 0; k < RS; ++k) A[k] = tab[k];
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
        int i = 0;
        for (; i + 1 < nobs; i += 2) {               // (two zero rows behind the table: the look-ahead never leaves it)
            const cptr nb = tab + (i + 1) * RS;
            // A is ready (waited for below / above); ask for B, compute A; wait, ask for the next A, compute B.  The waits
            // are explicit: scalar loads return out of order, so a wait is for ALL of them, and each row is asked for right
            // behind the wait that made its predecessor ready -- a whole observation (~220 cycles) ahead of its use.
            __builtin_amdgcn_sched_barrier(0);
#ifndef SMCN_ABL_LANE_NOROWS    // (ablation: every observation uses row 0 -- wrong values, the evaluation's time without scalar loads)
#pragma unroll
            for (int k = 0; k < RS; ++k) B[k] = nb[k];
#else
#pragma unroll
            for (int k = 0; k < RS; ++k) B[k] = A[k];
#endif
            __builtin_amdgcn_sched_barrier(0);
            rest(A, head(A));
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): B has landed
            __builtin_amdgcn_sched_barrier(0);
#ifndef SMCN_ABL_LANE_NOROWS
#pragma unroll
            for (int k = 0; k < RS; ++k) A[k] = nb[RS + k];
#endif
            __builtin_amdgcn_sched_barrier(0);
            rest(B, head(B));
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the next A has landed
            __builtin_amdgcn_sched_barrier(0);
        }
        if (i < nobs) rest(A, head(A));              // (an odd count's last observation)
        }
        ll -= lgsum;
        if (!(mumax < kInf) || mmin == 0.0) ll = -kInf;            // poisson_lpmf: lambda = inf; lambda = 0 with n != 0
        double lp = 0.0, dg = 0.0;
        gp[0] = 0.0;
#pragma unroll
        for (int c = 1; c < M; ++c) {                              // exponential-power priors of Beta_2..Beta_M (PRMwCD.stan:36-38)
            const double ab = fabs(x[c]);
            const double apm1 = rsqrt_nr(ab), apow = ab == 0.0 ? 0.0 : ab * apm1;
            const double p = apow * egq;
            lp += -g - p;
            dg += -1.0 + 0.5 * p;
            const double sgn = (x[c] > 0.0) ? 1.0 : ((x[c] < 0.0) ? -1.0 : 0.0);
            gp[c] = -0.5 * sgn * apm1 * egq;
        }
        lp += 2.0 * 0.26236426446749105203 - 3.0 * g - 1.3 * eg + g;   // inv_gamma(Gamma | 2, 1.3) + Jacobian
        dg += -3.0 + 1.3 * eg + 1.0;
        gp[M] = dg;
#pragma unroll
        for (int j = 0; j < M; ++j) gl[j] = acc[j];
        gl[M] = 0.0;
        lpri = lp;
        llik = ll;
    }
};


}  // namespace smcn
