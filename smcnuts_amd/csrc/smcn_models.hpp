// Device-native target densities (value + gradient in one pass), evaluated
// cooperatively by the G lanes that own a particle.
//
// Replaces StanModel.logpdf / logpdfgrad (smcnuts/model/bridgestan.py:28-90),
// whose BridgeStan back end is host-only.  Math restated from the .stan text:
// stan_models/arma/arma.stan:14-30, stan_models/PRMwCD/PRMwCD.stan:17-38.
//
// Every model returns the log prior (+ log-Jacobian) and the log likelihood
// separately (log pi_phi = lpri + phi * llik; arma.stan:30), and the two
// gradient parts.
//
// Model concept:
//   static constexpr int  G     lanes per particle
//   static constexpr int  DL    coordinates held per lane
//   static constexpr bool DIST  true: coordinate c = lg + G*i lives on lane lg
//                               false: every lane holds all DL = D coordinates
//   static constexpr int SHARED   doubles of block-shared LDS the model wants
//   static constexpr int MIN_WAVES  waves per SIMD the NUTS kernel is compiled for (register budget)
//   static constexpr int LDS_LEVELS tree-stack levels kept in LDS when the stack itself lives in HBM
//   int  dim()
//   void init(const double* mdata, int lg, double* shared)   all threads of the block
//   void eval(x[DL], lpri, llik, gpri[DL], glik[DL])   all lanes of the group
#pragma once
#include "smcn_device.hpp"

namespace smcn {

// ---------------------------------------------------------------------------
// Gaussian family: prior N(0, s0^2 I), optional likelihood N(x | m 1, s1^2 I).
// mdata = [D, s0, has_lik, m, s1].  Coordinates are distributed over lanes.
// ---------------------------------------------------------------------------
template <int G_, int DL_, int LEVELS = 2, int WAVES = 2>
struct GaussModel {
    static constexpr int G = G_, DL = DL_, SHARED = 0, MIN_WAVES = WAVES, LDS_LEVELS = LEVELS;
    static constexpr bool DIST = true;
    int D;
    double inv0, inv1, m, c0, c1;
    bool has;
    bool valid[DL];

    __device__ int dim() const { return D; }
    __device__ void init(const double* md, int lg, double*) {
        D = (int)md[0];
        const double s0 = md[1], s1 = md[4];
        has = md[2] != 0.0;
        m = md[3];
        inv0 = 1.0 / (s0 * s0);
        inv1 = 1.0 / (s1 * s1);
        c0 = -D * log(s0) - 0.5 * D * kLog2Pi;
        c1 = -D * log(s1) - 0.5 * D * kLog2Pi;
#pragma unroll
        for (int i = 0; i < DL; ++i) valid[i] = (lg + G * i) < D;
    }
    // the evaluation in two halves, so that a caller that owns the whole wavefront (G = 64) can put the two sums of
    // squares through ONE butterfly together with its own (the kinetic energy): this lane's share, then the totals
    static constexpr bool HAS_PARTIAL = true;
    __device__ __forceinline__ void eval_partial(const double (&x)[DL], double& ss, double& sl, double (&gp)[DL],
                                                 double (&gl)[DL]) const {
        ss = 0.0; sl = 0.0;
#pragma unroll
        for (int i = 0; i < DL; ++i) {
            const double xi = valid[i] ? x[i] : 0.0;
            const double d = valid[i] ? (x[i] - m) : 0.0;
            ss = fma(xi, xi, ss);
            sl = fma(d, d, sl);
            gp[i] = -xi * inv0;
            gl[i] = has ? -d * inv1 : 0.0;
        }
    }
    // nuts_wave_kernel (smcn_nuts_wave.hpp: one wavefront per particle, candidates by leaf index) is this model's NUTS kernel
    static constexpr bool WAVE_KERNEL = G_ == 64 && DL_ >= 2 && (DL_ % 2) == 0;
    // eval_partial with the two facts of the data that cost selects as compile-time constants: FULL (D = G * DL: every
    // slot is a coordinate) and HAS (there is a likelihood factor).  The same operations in the same order.
    template <bool FULL, bool HAS>
    __device__ __forceinline__ void eval_partial_t(const double (&x)[DL], double& ss, double& sl, double (&gp)[DL],
                                                   double (&gl)[DL]) const {
        ss = 0.0; sl = 0.0;
#pragma unroll
        for (int i = 0; i < DL; ++i) {
            const double xi = (FULL || valid[i]) ? x[i] : 0.0;
            ss = fma(xi, xi, ss);
            gp[i] = -xi * inv0;
            if constexpr (HAS) {
                const double d = (FULL || valid[i]) ? (x[i] - m) : 0.0;
                sl = fma(d, d, sl);
                gl[i] = -d * inv1;
            } else {
                gl[i] = 0.0;
            }
        }
    }
    __device__ __forceinline__ void finish(double ss_total, double sl_total, double& lpri, double& llik) const {
        lpri = -0.5 * ss_total * inv0 + c0;
        llik = has ? -0.5 * sl_total * inv1 + c1 : 0.0;
    }
    __device__ void eval(const double (&x)[DL], double& lpri, double& llik, double (&gp)[DL],
                         double (&gl)[DL]) const {
        double ss, sl;
        eval_partial(x, ss, sl, gp, gl);
        ss = group_sum<G>(ss);
        if (has) sl = group_sum<G>(sl);
        finish(ss, sl, lpri, llik);
    }
};

// ---------------------------------------------------------------------------
// PRMwCD with the particle state DISTRIBUTED over the group (coordinate c on lane c % G): the tree
// state of the NUTS kernel is then DL = ceil(13 / G) doubles per vector and lane instead of 13
// (the replicated form above needs > 256 VGPRs and spills).  One evaluation:
//   1. every lane publishes its coordinates in the group's LDS scratch and reads all 13 back,
//   2. lane lg accumulates the likelihood partials of observations lg, lg + G, ..,
//   3. the 13 gradient partials per lane go through the scratch again and lane c % G sums column c
//      (a reduce-scatter: G reads per owned coordinate instead of 13 butterflies),
//   4. prior terms are computed by the lane that owns the coordinate.
// The scratch is private to a group, whose lanes sit in one wavefront: LDS operations of a wave
// execute in order, so a wave barrier (no s_barrier) orders the exchange.
// ---------------------------------------------------------------------------
template <int G_, int NOBS, int C_, int RED = 0, int LEVELS = 2, bool FAST = false, int WAVES = 2>
struct PrmwcdDistModel {
    // FAST (round 4; the shipped shape only -- other data take the generic loop): the observation loop unrolled over the
    // lane's S observations with everything that is not arithmetic taken out of it.  The generic loop issues 101
    // instructions per observation for 47 of arithmetic (tools/ubench/prm_eval: 458 cycles per observation, and the
    // launch of BASELINE config 4 is its longest tree's leaf latency): a v_mov_b64 in front of nine of the
    // twelve Horner steps of exp (constants in VGPRs: v_fmac needs its addend in the destination), seventeen selects and
    // compares for `live` and for poisson_lpmf's two edge cases, two branches around the y loads, address arithmetic.
    // Here: design rows padded to S * G_ (all lanes live except in the last pass, whose padded lanes are masked), y in
    // the row's free slot, rows at immediate offsets from one per-lane base, exp's constants as scalar operands
    // (v_fma_f64 with an SGPR pair), and the edge cases decided ONCE behind the loop from max(mu) and
    // min(mu + [y == 0]).  Same operations in the same order for every sum: bit-identical results.
    // RED: how the gradient partials are reduce-scattered: 0 = through G rows of LDS scratch,
    //      1 = two DPP stages first, then 2 rows (G = 8), 2 = DPP only (no scratch at all)
    static constexpr int LDS_LEVELS = LEVELS;             // tree-stack levels kept in LDS (hybrid stack)
    static constexpr int G = G_, C = C_, M = C_ + 1, D_ = C_ + 2, DL = (C_ + 2 + G_ - 1) / G_;
    static constexpr int RS = (C_ + 1 + 1) & ~1;          // design row, padded to an even count
    static constexpr int PR = (D_ + 2) & ~1;              // partial row: 13 -> 14 doubles
    static constexpr int SCR = RED == 0 ? G_ * PR : (RED == 1 ? (G_ / 4) * PR : 0);   // per-group exchange scratch
    static constexpr int SG = ((NOBS + G_ - 1) / G_) * G_;                   // observations padded to whole passes
    static constexpr int XROWS = FAST ? SG : NOBS;
    static constexpr int DATA = XROWS * RS + 2 * NOBS + (FAST ? 2 * SG : 0);   // design, y, lgamma(y + 1) (+ FAST: [lgamma, y == 0] pairs)
    static constexpr int SHARED = ((DATA + 1) & ~1) + (256 / G_) * SCR, MIN_WAVES = WAVES;
    static constexpr bool DIST = true;
    // one wavefront per particle (the kernel that finishes parked trees: smcn_set_nuts_cap): its whole tree stack would
    // fit LDS, but the edges are to live in registers as in the kernel that parked the tree -- the hybrid-stack form of
    // nuts_kernel with every level in LDS (LEVELS = 10) and an HBM slot nothing ever touches
    static constexpr bool HYBRID_ALWAYS = G_ >= 64;
    static constexpr bool TWO_PHASE = true;               // nuts_kernel: park / resume at a doubling boundary
    static constexpr bool FIN_KERNEL = G_ >= 64;          // nuts_fin_kernel (smcn_nuts_fin.hpp) finishes / builds this functor's trees
    // nuts_kernel: its groups (8 particles a wavefront) take a new particle only every 16th loop iteration -- trees of
    // hundreds of leaves lose nothing by the wait, and the groups then do their deep merges in the same iterations
    // (config 4: 1.47 -> 1.52 G leapfrog/s; 2 / 4 / 8 / 16 / 32 / 64 within 1 % of each other)
    static constexpr int STEP_ALIGN = 16;
    static constexpr int S = (NOBS + G - 1) / G;
    // NOBS and C_ are CAPACITIES: the data's own shape (nobs <= NOBS observations, cc <= C_ kernel columns, the
    // shipped file: 100 and 11) is read from mdata; unused design entries are zero, unused coordinates masked
    int lg, nobs, cc;
    double q;
    const double* X;  // [NOBS][RS] in LDS
    const double* y;  // [NOBS]     in LDS
    double* scr;      // [G][PR]    in LDS, this group's

    __device__ int dim() const { return cc + 2; }
    __device__ void init(const double* md, int lg_, double* shared) {
        lg = lg_;
        nobs = (int)md[0];
        cc = (int)md[2];
        q = md[3];
        for (int t = threadIdx.x; t < XROWS * RS; t += blockDim.x) {
            const int i = t / RS, j = t - i * RS;
            double v = (i < nobs && j < cc) ? md[4 + nobs + i * cc + j] : 0.0;
            if (FAST && j == RS - 1) v = i < nobs ? md[4 + i] : 0.0;          // y_i rides in the row's padding slot
            shared[t] = v;
        }
        for (int t = threadIdx.x; t < NOBS; t += blockDim.x) {
            shared[XROWS * RS + t] = t < nobs ? md[4 + t] : 0.0;
            shared[XROWS * RS + NOBS + t] = t < nobs ? lgamma(md[4 + t] + 1.0) : 0.0;   // data-only term of poisson_lpmf
        }
        if constexpr (FAST) {
            static_assert(!FAST || RS > C_, "FAST needs a free slot in the design row (odd column count)");
            for (int t = threadIdx.x; t < SG; t += blockDim.x) {
                shared[XROWS * RS + 2 * NOBS + 2 * t] = t < nobs ? lgamma(md[4 + t] + 1.0) : 0.0;
                shared[XROWS * RS + 2 * NOBS + 2 * t + 1] = (t < nobs && md[4 + t] != 0.0) ? 0.0 : 1.0;
            }
        }
        X = shared;
        y = shared + XROWS * RS;
        scr = shared + ((DATA + 1) & ~1) + (threadIdx.x / G) * SCR;
        __syncthreads();
    }
    static __device__ __forceinline__ void wave_sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // One wavefront per particle (G_ = 64, FAST; the kernel that finishes the long trees a two-phase launch parks): the
    // evaluation built for LATENCY.  Coordinate c lives on lane c; the 13 coefficients are read out as scalars
    // (v_readlane) and enter the FMAs as SGPR operands; lane l evaluates observations l and l + 64 with the unrolled
    // loop's per-observation code; the 12 gradient sums, the log-likelihood and the two prior sums go through four
    // four-value butterflies (wave_sum4) and come back as scalars, which lane c picks its own from.  ~370 wave
    // instructions where the generic evaluation on 64 lanes issued ~1 000.  (The sums are associated differently
    // from the 8-lane kernel's: the same values to rounding -- the short-tree parity tests run with this functor, too -- and, on
    // this target's chaotic trajectories, another equally valid tree: DESIGN.md 2.)
    __device__ bool eval_wave(const double (&x)[DL], double& lpri, double& llik, double (&gp)[DL], double (&gl)[DL]) const {
        if constexpr (G_ == 64 && FAST && DL == 1) {
            if (!(nobs > 64 && nobs <= 2 * 64 && cc == C && q == 0.5)) return false;
            using d2 = double __attribute__((ext_vector_type(2)));
            using lds2 = const __attribute__((address_space(3))) d2*;
            double bs[D_];
#pragma unroll
            for (int j = 0; j < D_; ++j) bs[j] = lane_value(x[0], j);
            const double g = bs[M];
            const double egq = exp_fast(-0.5 * g), eg = egq * egq;
            const lds2 rows = (lds2)(X + lg * RS);
            const lds2 lz = (lds2)(y + 2 * NOBS) + lg;
            double pa[M], pll = 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) pa[j] = 0.0;
            bool edge = false;
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                double row[RS];
#pragma unroll
                for (int j2 = 0; j2 < RS / 2; ++j2) {
                    const d2 t = rows[(o * 64 * RS) / 2 + j2];
                    row[2 * j2] = t.x; row[2 * j2 + 1] = t.y;
                }
                const d2 aux = lz[o * 64];
                double e = bs[0];
#pragma unroll
                for (int j = 0; j < C; ++j) e = fma(bs[j + 1], row[j], e);
                const double mu = exp_fast_s(e);
                const double yi = row[RS - 1];
                double t1, term;
                {
#pragma clang fp contract(off)
                    t1 = yi * e;
                    term = (t1 - mu) - aux.x;
                }
                double d = yi - mu;
                bool bad = !(mu < kInf) || (mu + aux.y) == 0.0;         // poisson_lpmf: lambda = inf; lambda = 0 with n != 0
                if (o == 1) {                                            // lanes past the data
                    const bool live = lg + 64 < nobs;
                    term = live ? term : 0.0; d = live ? d : 0.0; bad = live && bad;
                }
                edge = edge || bad;
                pll += term;
                pa[0] += d;
#pragma unroll
                for (int j = 0; j < C; ++j) pa[j + 1] = fma(d, row[j], pa[j + 1]);
            }
            // priors on the owning lane (PRMwCD.stan:21, 36-38), as in eval()
            double plp = 0.0, pdg = 0.0, gpl = 0.0;
            if (lg >= 1 && lg < M) {
                const double ab = fabs(x[0]);
                const double apm1 = rsqrt_nr(ab), apow = ab == 0.0 ? 0.0 : ab * apm1;
                const double p = apow * egq;
                plp = -g - p;
                pdg = -1.0 + 0.5 * p;
                const double sgn = (x[0] > 0.0) ? 1.0 : ((x[0] < 0.0) ? -1.0 : 0.0);
                gpl = -0.5 * sgn * apm1 * egq;
            } else if (lg == M) {
                plp = 2.0 * 0.26236426446749105203 - 3.0 * g - 1.3 * eg + g;
                pdg = -3.0 + 1.3 * eg + 1.0;
            }
            double S[16];
            wave_sum4(pa[0], pa[1], pa[2], pa[3], S[0], S[1], S[2], S[3]);
            wave_sum4(pa[4], pa[5], pa[6], pa[7], S[4], S[5], S[6], S[7]);
            wave_sum4(pa[8], pa[9], pa[10], pa[11], S[8], S[9], S[10], S[11]);
            wave_sum4(pll, plp, pdg, 0.0, S[12], S[13], S[14], S[15]);
            double glv = 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) glv = (lg == j) ? S[j] : glv;
            gl[0] = glv;
            gp[0] = (lg == M) ? S[14] : gpl;
            lpri = S[13];
            llik = (__ballot(edge) != 0ull) ? -kInf : S[12];
            return true;
        }
        return false;
    }

    __device__ void eval(const double (&x)[DL], double& lpri, double& llik, double (&gp)[DL],
                         double (&gl)[DL]) const {
        if constexpr (G_ == 64 && FAST && DL == 1) {
            if (eval_wave(x, lpri, llik, gp, gl)) return;
        }
        // ---- 1. all coordinates to every lane
        double b[PR];
        if constexpr (RED == 2 && G_ == 4) {                      // four lanes per particle: a quad broadcast, no LDS crossbar
#pragma unroll
            for (int j = 0; j < PR; ++j) b[j] = (j < D_) ? quad_read(x[j / G], j % G) : 0.0;
        } else if constexpr (RED == 2) {
#pragma unroll
            for (int j = 0; j < PR; ++j) b[j] = (j < D_) ? group_read<G>(x[j / G], j % G) : 0.0;
        } else {
#pragma unroll
            for (int i = 0; i < DL; ++i)
                if (lg + G * i < PR) scr[lg + G * i] = (lg + G * i < D_) ? x[i] : 0.0;
            wave_sync();
#pragma unroll
            for (int j = 0; j < PR; ++j) b[j] = scr[j];
            wave_sync();
        }
        const int Mr = cc + 1;                             // index of g = log Gamma
        double g = 0.0;
        if (FAST && Mr == M) g = b[M];                     // (the shipped shape: no 13-way select)
        else {
#pragma unroll
            for (int j = 1; j < PR; ++j) g = (j == Mr) ? b[j] : g;
        }
        if (Mr != M) {                                     // (fewer columns than the capacity: g must not meet a zero
#pragma unroll                                             //  design entry as inf * 0)
            for (int j = 1; j < PR; ++j) b[j] = (j < Mr) ? b[j] : 0.0;
        }
        const bool half = q == 0.5;                        // the shipped data; a branch, not a select
        double eg, egq;                                    // 1 / Gamma, Gamma^-q
        if (half) { egq = FAST ? exp_fast_s(-0.5 * g) : exp_fast(-0.5 * g); eg = egq * egq; }
        else { eg = exp(-g); egq = pow(eg, q); }
        // ---- 2. likelihood partials of this lane's observations (PRMwCD.stan:24-33)
        double ll = 0.0, acc[PR];
#pragma unroll
        for (int j = 0; j < PR; ++j) acc[j] = 0.0;
        const int Sr = (nobs + G - 1) / G;
        if (FAST && Sr == S && Mr == M) {      // (the shipped shape; other data take the loop below)
            using d2 = double __attribute__((ext_vector_type(2)));
            using lds2 = const __attribute__((address_space(3))) d2*;
            const lds2 rows = (lds2)(X + lg * RS);                            // this lane's first row; the next G * RS doubles on
            const lds2 lz = (lds2)(y + 2 * NOBS) + lg;                        // [lgamma(y + 1), y == 0] of its observations
            double mumax = 0.0, mmin = 1.0;
            auto one = [&](int k, bool masked) __attribute__((always_inline)) {
                double row[RS];
#pragma unroll
                for (int j2 = 0; j2 < RS / 2; ++j2) {
                    const d2 t = rows[(k * G * RS) / 2 + j2];
                    row[2 * j2] = t.x; row[2 * j2 + 1] = t.y;
                }
                const d2 aux = lz[k * G];
                double e = b[0];
#pragma unroll
                for (int j = 0; j < C; ++j) e = fma(b[j + 1], row[j], e);
                const double mu = exp_fast_s(e);
                const double yi = row[RS - 1];
                double t1, term;
                {
#pragma clang fp contract(off)
                    t1 = yi * e;                                             // (0 for y = 0, as the reference's select)
                    term = (t1 - mu) - aux.x;
                }
                double d = yi - mu, mz = mu + aux.y, mm = mu;
                if (masked) {                                                // the last pass: lanes past the data
                    const bool live = lg + G * k < nobs;
                    term = live ? term : 0.0; d = live ? d : 0.0; mz = live ? mz : 1.0; mm = live ? mm : 0.0;
                }
                mumax = fmax(mumax, mm);
                mmin = fmin(mmin, mz);
                ll += term;
                acc[0] += d;
#pragma unroll
                for (int j = 0; j < C; ++j) acc[j + 1] = fma(d, row[j], acc[j + 1]);
            };
#pragma unroll
            for (int k = 0; k < S; ++k) {
                one(k, k == S - 1 && SG != NOBS);
                if (k & 1) __builtin_amdgcn_sched_barrier(0);                // two observations in flight, not thirteen rows
            }
            // poisson_lpmf's edge cases (lambda = inf; lambda = 0 with n != 0), once: -inf as the reference
            if (!(mumax < kInf) || mmin == 0.0) ll = -kInf;
        } else
#pragma unroll 1
        for (int k = 0; k < Sr; ++k) {
            const int i = lg + G * k;
            const bool live = i < nobs;
            const double* row = X + (live ? i : 0) * RS;
            double eta = b[0];
#pragma unroll
            for (int j = 0; j < C; ++j) eta = fma(b[j + 1], row[j], eta);
            const double mu = exp_fast(eta);
            const double yi = live ? y[i] : 0.0;
            double term = (yi == 0.0 ? 0.0 : yi * eta) - mu - (live ? y[NOBS + i] : 0.0);
            term = (mu == 0.0 && yi != 0.0) ? -kInf : term;              // lambda == 0, n != 0
            term = finite_d(mu) ? term : -kInf;                          // poisson_lpmf(y | inf)
            const double d = live ? (yi - mu) : 0.0;
            ll += live ? term : 0.0;
            acc[0] += d;
#pragma unroll
            for (int j = 0; j < C; ++j) acc[j + 1] = fma(d, row[j], acc[j + 1]);
        }
        // ---- 3. reduce-scatter of the gradient partials: lane c % G ends with the sum of column c
        if constexpr (RED == 0) {
#pragma unroll
            for (int j = 0; j < PR; ++j) scr[lg * PR + j] = acc[j];
            wave_sync();
#pragma unroll
            for (int i = 0; i < DL; ++i) {
                const int c = lg + G * i;
                double sum = 0.0;
                if (c < PR) {
#pragma unroll
                    for (int l = 0; l < G; ++l) sum += scr[l * PR + c];
                }
                gl[i] = (c < D_) ? sum : 0.0;
            }
            wave_sync();
        } else if constexpr (RED == 1) {
            static_assert(RED != 1 || G >= 4, "RED = 1 needs quads");
#pragma unroll
            for (int j = 0; j < D_; ++j) {
                acc[j] += dpp_mov<0xB1>(acc[j]);
                acc[j] += dpp_mov<0x4E>(acc[j]);       // every lane of a quad holds the quad's sum
            }
#pragma unroll
            for (int j = 0; j < D_; ++j)
                if ((j & 3) == (lg & 3)) scr[(lg >> 2) * PR + j] = acc[j];
            wave_sync();
#pragma unroll
            for (int i = 0; i < DL; ++i) {
                const int c = lg + G * i;
                double sum = 0.0;
                if (c < D_) {
#pragma unroll
                    for (int l = 0; l < G / 4; ++l) sum += scr[l * PR + c];
                }
                gl[i] = sum;
            }
            wave_sync();
        } else {
#pragma unroll
            for (int j = 0; j < D_; ++j) acc[j] = group_sum<G>(acc[j]);
#pragma unroll
            for (int i = 0; i < DL; ++i) {
                double v = 0.0;
#pragma unroll
                for (int j = i * G; j < (i + 1) * G && j < D_; ++j) v = (j - i * G == lg) ? acc[j] : v;
                gl[i] = v;
            }
        }
        // ---- 4. priors on the owning lane: inv_gamma(Gamma | 2, 1.3) + Jacobian for g,
        //         exponential-power terms for Beta_2..Beta_M (:36-38); Beta_1 is flat
        double lp = 0.0, dg = 0.0;
#pragma unroll
        for (int i = 0; i < DL; ++i) {
            const int c = lg + G * i;
            gp[i] = 0.0;
            if (c >= 1 && c < Mr) {
                const double ab = fabs(x[i]);
                double apow, apm1;                                         // |Beta_j|^q, |Beta_j|^(q-1)
                if (half) { apm1 = rsqrt_nr(ab); apow = ab == 0.0 ? 0.0 : ab * apm1; }
                else { apow = pow(ab, q); apm1 = apow / ab; }
                const double p = apow * egq;                               // (|Beta_j| / Gamma)^q
                lp += -g - p;
                dg += -1.0 + q * p;
                const double sgn = (x[i] > 0.0) ? 1.0 : ((x[i] < 0.0) ? -1.0 : 0.0);
                gp[i] = -q * sgn * apm1 * egq;                             // -q sgn |b|^(q-1) e^(-gq)
            } else if (c == Mr) {
                lp += 2.0 * 0.26236426446749105203 - 3.0 * g - 1.3 * eg + g;   // lgamma(2) = 0
                dg += -3.0 + 1.3 * eg + 1.0;
            }
        }
        llik = group_sum<G>(ll);
        lpri = group_sum<G>(lp);
        dg = group_sum<G>(dg);
#pragma unroll
        for (int i = 0; i < DL; ++i)
            if (lg + G * i == Mr) gp[i] = dg;
    }
};


}  // namespace smcn
