// The NUTS proposal for targets whose particle fills a wavefront (Gaussians of 65..512 dimensions: BASELINE
// config 5, D = 256): ONE WAVEFRONT OWNS ONE PARTICLE, coordinate c = lane + 64 k on lane `lane`.
//
// Replaces NUTSProposal.rvs / generate_nuts_samples / build_tree / NUTSLeapfrog / stop_criterion
// (smcnuts/proposal/nuts.py:34-175) like nuts_kernel (smcn_nuts.hpp), whose per-leaf state machine this is -- same
// draws at the same places, same merges in the same order (nuts.py:134-148) -- with two differences that follow
// from the mapping:
//
//  * The whole tree control (leaf and doubling counters, slice variable, sub-tree weights, stop flags) is the same in
//    every lane.  It is written as wave-uniform values from the start -- conditions come back through a ballot, sums
//    through v_readlane -- so that every branch is a scalar branch and no vector is ever moved by a select.
//
//  * CANDIDATES BY LEAF INDEX.  build_tree carries a candidate (x', r') up the recursion, which it needs BY VALUE once:
//    when the tree has ended (nuts.py:93-101 keeps the top-level one, :142-144 the sub-trees').  Until then a candidate
//    is fully named by the signed index of its leaf along the trajectory (+k: the k-th leapfrog forward of the start,
//    -k: backward).  So every stack level holds two integers (index, weight n') in a lane of a register instead of
//    2 D + 2 doubles in LDS / HBM, a merge swaps an integer, and at the tree's end the selected leaf is re-integrated from
//    (x0, r0) by |index| bare leapfrogs (two FMAs and the gradient per coordinate: no reductions, no tests).  Each
//    direction's chain is the same sequence of operations whenever it is run, so the replayed leaf is bit for bit the
//    leaf the tree visited.  What is left of the tree stack is the FIRST LEAF of every open sub-tree (the U-turn test
//    of its merge, nuts.py:148, needs both ends): 2 D doubles per level, levels 1..4 in LDS (15/16 of all accesses),
//    the rest in an HBM slot of the resident wavefront.
//
// The trajectory's two edges live in registers as (x, r, grad) of the MOVING edge -- which is the live state of the
// integrator -- and of the parked one; a change of direction swaps them.
#pragma once
#include "smcn_nuts.hpp"

namespace smcn {

#ifndef SMCN_WAVE_R0
#define SMCN_WAVE_R0 2        // A/B: the start momentum comes back from global memory (0), from an LDS slot (1), from registers (2)
#endif
constexpr int kWaveLdsSlots = 4;   // LDS slots of a wavefront (4 KB each at 4 coordinates per lane): the start momentum + first leaves 1..3

template <class M, class = void>
struct model_wave_kernel { static constexpr bool value = false; };
template <class M>
struct model_wave_kernel<M, std::enable_if_t<M::WAVE_KERNEL>> { static constexpr bool value = true; };

__host__ __device__ constexpr int wave_slot_doubles(int DL) { return 2 * DL * 64; }

// a wave-uniform condition as a scalar (all lanes are active in this kernel: the control flow is uniform)
__device__ __forceinline__ bool wuni(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }
__device__ __forceinline__ int64_t wfirst64(int64_t v) {
    return (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)v) | ((int64_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32);
}
__device__ __forceinline__ double wfirst(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// FULL: every lane's DL coordinates are real (D = 64 DL: no masking anywhere); HAS: the target has a likelihood factor
// (both are facts of the model data the launcher knows: compile-time here, so that neither costs a select per coordinate)
// SLOTS: LDS slots per wavefront (slot 0: the start momentum r0, which the replay needs again; slots 1..: first leaves);
// WAVES: wavefronts per SIMD the kernel is compiled for.
template <class Model, bool FULL, bool HAS, int SLOTS = kWaveLdsSlots, int WAVES = Model::MIN_WAVES>
__global__ void __launch_bounds__(kNutsBlock, WAVES) nuts_wave_kernel(NutsArgs a) {
    static_assert(Model::G == 64 && Model::DIST && (Model::DL % 2) == 0, "one wavefront per particle, pairs of coordinates");
    constexpr int R0S = SMCN_WAVE_R0 == 1 ? 1 : 0;
    constexpr int DL = Model::DL, LF = SLOTS - R0S, SLOTD = wave_slot_doubles(DL);
    using d2 = double __attribute__((ext_vector_type(2)));
    using lds2 = __attribute__((address_space(3))) d2*;
    using glb2 = __attribute__((address_space(1))) d2*;

    extern __shared__ double lds[];
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = (int)(threadIdx.x >> 6);
    // first leaves: [slot][pair][lane] pairs of doubles -- x pairs first, then r pairs; every access a conflict-free b128
    const lds2 r0s = (lds2)(lds + wave * SLOTS * SLOTD) + lane;            // slot 0: r0 in its first DL / 2 pairs
    const lds2 fl = r0s + R0S * (SLOTD / 2);
    const glb2 fg = (glb2)(a.scratch + ((int64_t)blockIdx.x * (kNutsBlock / 64) + wave) * (int64_t)(kMaxLevels * SLOTD)) + lane;

    auto kargs = [&]() __attribute__((always_inline)) {     // per-tree pointers: re-read where used (smcn_nuts.hpp)
        using kptr = const __attribute__((address_space(4))) NutsArgs*;
        kptr kp = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        return kp;
    };
    Model model;
    model.init(a.mdata, lane, lds);
    const int D = model.dim();
    const int64_t N = a.N;
    const double eps = a.eps, phi = a.phi;
    bool cv[DL];
    int64_t cidx[DL];
#pragma unroll
    for (int k = 0; k < DL; ++k) {
        const int c = lane + 64 * k;
        cv[k] = c < D;
        cidx[k] = (int64_t)c * N;
    }

    auto store_first = [&](int s, const double (&x)[DL], const double (&r)[DL]) {      // s: wave-uniform slot
        if (s < LF) {
            const lds2 p = fl + s * (SLOTD / 2);
#pragma unroll
            for (int t = 0; t < DL / 2; ++t) {
                p[t * 64] = d2{x[2 * t], x[2 * t + 1]};
                p[(DL / 2 + t) * 64] = d2{r[2 * t], r[2 * t + 1]};
            }
        } else {
            const glb2 p = fg + (int64_t)s * (SLOTD / 2);
#pragma unroll
            for (int t = 0; t < DL / 2; ++t) {
                p[t * 64] = d2{x[2 * t], x[2 * t + 1]};
                p[(DL / 2 + t) * 64] = d2{r[2 * t], r[2 * t + 1]};
            }
        }
    };
    auto load_first = [&](int s, double (&x)[DL], double (&r)[DL]) {
        if (s < LF) {
            const lds2 p = fl + s * (SLOTD / 2);
#pragma unroll
            for (int t = 0; t < DL / 2; ++t) {
                const d2 u = p[t * 64], v = p[(DL / 2 + t) * 64];
                x[2 * t] = u.x; x[2 * t + 1] = u.y; r[2 * t] = v.x; r[2 * t + 1] = v.y;
            }
        } else {
            const glb2 p = fg + (int64_t)s * (SLOTD / 2);
#pragma unroll
            for (int t = 0; t < DL / 2; ++t) {
                const d2 u = p[t * 64], v = p[(DL / 2 + t) * 64];
                x[2 * t] = u.x; x[2 * t + 1] = u.y; r[2 * t] = v.x; r[2 * t + 1] = v.y;
            }
        }
    };
    // nuts.py:152-160 between the trajectory ends (xm, rm) and (xp, rp)
    auto uturn = [&](const double (&xm)[DL], const double (&rm)[DL], const double (&xp)[DL], const double (&rp)[DL]) -> bool {
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int k = 0; k < DL; ++k) {
            const double dx = xp[k] - xm[k];
            sa = __builtin_fma(dx, rm[k], sa);
            sb = __builtin_fma(dx, rp[k], sb);
        }
        wave_sum2(sa, sb, sa, sb);
        return wuni((sa < 0.0) || (sb < 0.0));
    };

    // ---- draws: Philox blocks of 128 (two per lane), or the recorded tape (tests) ---------------------------------
    int64_t p = 0;
    uint32_t q = 0, qbase = 0;
    double ub0 = 0.0, ub1 = 0.0;
    int64_t toff = 0, tlen = 0;
    bool overflow = false;
    const bool taped = a.tape != nullptr;
    auto refill = [&]() {
        const u32x4 o = philox4x32_10({(qbase >> 1) + (uint32_t)lane, (uint32_t)(a.particle_base + p), a.iter, kStreamNuts},
                                      (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
        ub0 = u53(o.a, o.b);
        ub1 = u53(o.c, o.d);
    };
    auto draw = [&]() -> double {
        double v;
        if (taped) {
            if ((int64_t)q < tlen) v = wfirst(a.tape[toff + q]);
            else { v = 0.5; overflow = true; }
        } else {
            if (q >= qbase + 128u) { qbase += 128u; refill(); }
            const int src = (int)((q - qbase) >> 1);
            if (q & 1u) v = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ub1), src), __builtin_amdgcn_readlane(__double2loint(ub1), src));
            else v = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ub0), src), __builtin_amdgcn_readlane(__double2loint(ub0), src));
        }
        ++q;
        return v;
    };

    // gradient of log pi_phi from its two parts (without a likelihood the second part is 0: phi * 0 + gp = gp up to the
    // sign of a zero); coordinates past D stay 0
    auto grad = [&](double gp, double gl, int k) -> double {
        const double v = HAS ? __builtin_fma(phi, gl, gp) : gp;
        return (FULL || cv[k]) ? v : 0.0;
    };
    // one leapfrog WITHOUT value, kinetic energy or tests (nuts.py:169-173): the replay of the selected leaf's chain
    auto bare_step = [&](double (&x)[DL], double (&r)[DL], double (&g)[DL], double e, double h) {
#pragma unroll
        for (int k = 0; k < DL; ++k) r[k] = __builtin_fma(h, g[k], r[k]);
#pragma unroll
        for (int k = 0; k < DL; ++k) x[k] = __builtin_fma(e, r[k], x[k]);
        double ss, sl, gp[DL], gl[DL];
        model.template eval_partial_t<FULL, HAS>(x, ss, sl, gp, gl);
#pragma unroll
        for (int k = 0; k < DL; ++k) {
            g[k] = grad(gp[k], gl[k], k);
            r[k] = __builtin_fma(h, g[k], r[k]);
        }
    };
    // one leapfrog with value and |r'|^2 (one four-value butterfly).  A non-finite density -- the rare case in which the
    // reference's adapter overrides the gradient (bridgestan.py:79-80) -- takes the plain path of nuts_kernel.
    auto full_step = [&](double (&x)[DL], double (&r)[DL], double (&g)[DL], double e, double h, double& lpri, double& llik,
                         double& lp, double& kin) {
#pragma unroll
        for (int k = 0; k < DL; ++k) r[k] = __builtin_fma(h, g[k], r[k]);
#pragma unroll
        for (int k = 0; k < DL; ++k) x[k] = __builtin_fma(e, r[k], x[k]);
        double ss, sl, gp[DL], gl[DL], rk[DL], kp = 0.0;
        model.template eval_partial_t<FULL, HAS>(x, ss, sl, gp, gl);
#pragma unroll
        for (int k = 0; k < DL; ++k) {
            g[k] = grad(gp[k], gl[k], k);
            rk[k] = __builtin_fma(h, g[k], r[k]);
            kp = __builtin_fma(rk[k], rk[k], kp);
        }
        double sst, slt, unused;
        wave_sum4(ss, sl, kp, 0.0, sst, slt, kin, unused);
        model.finish(sst, slt, lpri, llik);
        lp = lpri + phi * llik;
        if (wuni(finite_d(lp))) {
#pragma unroll
            for (int k = 0; k < DL; ++k) r[k] = rk[k];
        } else {
            lp = -kInf;
            double kq = 0.0;
#pragma unroll
            for (int k = 0; k < DL; ++k) {
                g[k] = (FULL || cv[k]) ? -kInf : 0.0;
                r[k] = __builtin_fma(h, g[k], r[k]);
                kq = __builtin_fma(r[k], r[k], kq);
            }
            kin = group_sum<64>(kq);
        }
    };

    // ---- work: lines of 8 particles dealt to the XCDs (smcn_nuts.hpp: one L2 per line), a ticket per particle, taken when
    // the wavefront is free.  (Measured and dropped, profiles/r05_c5_ab.txt: the ticket a whole tree ahead -- a reserved
    // particle is one no idle wavefront can take: the same instruction and wave-cycle counts, 13 % more GRBM_GUI_ACTIVE at
    // step 0.1 --; the ticket asked for before the replay and the next particle's loads in front of this tree's stores:
    // 17 % slower at step 0.25, where every tree has 15 leaves and the wavefronts move in phase.  What hides the round
    // trips at a tree's two ends is the third wavefront per SIMD.)
    const unsigned int nq = gridDim.x < 8u ? gridDim.x : 8u;
    const unsigned int xq = blockIdx.x % nq;
    for (;;) {
        int64_t pp = -1;
        for (;;) {
            unsigned int t = 0;
            if (lane == 0) t = atomicAdd(a.queue + 8 + xq, 1u);
            t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
            const int64_t line = (int64_t)(t >> 3) * nq + xq;
            if (line * 8 >= N) break;
            if (line * 8 + (t & 7u) < N) { pp = line * 8 + (t & 7u); break; }
        }
        if (pp < 0) break;
        p = pp;
        double x[DL], r[DL], g[DL], x0[DL];
        {
            const auto ka = kargs();
            const double* const xin = ka->x;
            const double* const rin = ka->r;
            const bool rpm = ka->r_pm != 0;            // the momentum as one contiguous row per particle (smcn_ctx::r_pm)
#pragma unroll
            for (int k = 0; k < DL; ++k) {
                x[k] = (FULL || cv[k]) ? xin[cidx[k] + p] : 0.0;
                r[k] = (FULL || cv[k]) ? (rpm ? rin[p * D + lane + 64 * k] : rin[cidx[k] + p]) : 0.0;
                x0[k] = x[k];
            }
        }
        [[maybe_unused]] double r0[DL];
        if constexpr (SMCN_WAVE_R0 == 1) {
#pragma unroll
            for (int t = 0; t < DL / 2; ++t) r0s[t * 64] = d2{r[2 * t], r[2 * t + 1]};
        } else if constexpr (SMCN_WAVE_R0 == 2) {
#pragma unroll
            for (int k = 0; k < DL; ++k) r0[k] = r[k];
        }
        q = 0; qbase = 0; overflow = false;
        if (taped) { toff = a.tape_off[p]; tlen = a.tape_off[p + 1] - toff; }
        else refill();
        // ---- start of the tree (nuts.py:66-87) ------------------------------------------------------------------------
        double lpri0, llik0, kin_start, logu;
        {
            double ss, sl, gp[DL], gl[DL], kp = 0.0, unused;
            model.template eval_partial_t<FULL, HAS>(x, ss, sl, gp, gl);
#pragma unroll
            for (int k = 0; k < DL; ++k) kp = __builtin_fma(r[k], r[k], kp);
            double sst, slt;
            wave_sum4(ss, sl, kp, 0.0, sst, slt, kin_start, unused);
            model.finish(sst, slt, lpri0, llik0);
            double lp = lpri0 + phi * llik0;
            const bool bad = !wuni(finite_d(lp));          // bridgestan.py:47-49,79-80
            lp = bad ? -kInf : lp;
#pragma unroll
            for (int k = 0; k < DL; ++k) g[k] = bad ? ((FULL || cv[k]) ? -kInf : 0.0) : grad(gp[k], gl[k], k);
            if (lane == 0) {
                const auto ka = kargs();
                ka->lpri0[p] = lpri0; ka->llik0[p] = llik0;
                if (ka->kin0) ka->kin0[p] = kin_start;
            }
            double ex = draw();
            if (!taped) ex = -log1p(-ex);
            logu = (lp - 0.5 * kin_start) - ex;            // H0 - Exp(1)
        }
        double px[DL], pr[DL], pg[DL];                     // the parked edge (the live state is the moving one)
#pragma unroll
        for (int k = 0; k < DL; ++k) { px[k] = x[k]; pr[k] = r[k]; pg[k] = g[k]; }
        int j = 0, n = 1, nleap = 0, sel = 0;
        int nfw = 0, nbw = 0;                              // leaves built forward / backward of the start so far
        int dir = (draw() < 0.5) ? 1 : -1;                 // nuts.py:91
        int lvl_n = 0, lvl_i = 0;                          // lane m: weight n' and candidate leaf of the pending first half of level m + 1
        bool stop = false;
        while (!stop) {
            // ---- one doubling: 2^j leaves in direction dir (nuts.py:93-96 -> build_tree) --------------------------------
            const double e = dir > 0 ? eps : -eps, h = dir > 0 ? 0.5 * eps : -0.5 * eps;
            const int base = dir > 0 ? nfw : nbw;
            int nsub = 0, cand = 0;
            bool ssub = false;
            int i = 0;
            for (;; ++i) {
                double lpri, llik, lp, kin;
                full_step(x, r, g, e, h, lpri, llik, lp, kin);
                ++nleap;
                const double joint = lp - 0.5 * kin;       // nuts.py:123-125
                nsub = wuni(logu < joint) ? 1 : 0;
                ssub = wuni((logu - a.delta_max) >= joint);
                cand = dir * (base + 1 + i);
                if (j > 0 && (i & 1) == 0) store_first(((i == 0) ? j : (__builtin_ctz((unsigned)i))) - 1, x, r);
                // ---- merge completed sub-trees (nuts.py:134-148) -------------------------------------------------------
                bool done = false;
                for (int m = 0;; ++m) {
                    if (m == j) { done = true; break; }
                    if (ssub) {
                        // the stop unwinds the recursion: each ancestor for which the stopped sub-tree is the SECOND half
                        // still consumes its merge uniform
                        q += (uint32_t)__builtin_popcount((unsigned)(i >> m) & ((1u << (j - m)) - 1u));
                        done = true;
                        break;
                    }
                    if (((i >> m) & 1) == 0) {
                        lvl_n = (lane == m) ? nsub : lvl_n;
                        lvl_i = (lane == m) ? cand : lvl_i;
                        break;
                    }
                    const double u = draw();               // nuts.py:142, always
                    const int n1 = __builtin_amdgcn_readlane(lvl_n, m);
                    const int den = (n1 + nsub) > 1 ? (n1 + nsub) : 1;
                    if (!wuni(u < (double)nsub / (double)den)) cand = __builtin_amdgcn_readlane(lvl_i, m);
                    nsub += n1;                            // :146
                    const int i0 = (i >> (m + 1)) << (m + 1);
                    double fx[DL], fr[DL];
                    load_first(((i0 == 0) ? j : __builtin_ctz((unsigned)i0)) - 1, fx, fr);
                    ssub = dir > 0 ? uturn(fx, fr, x, r) : uturn(x, r, fx, fr);      // :148
                }
                if (done) break;
            }
            // ---- end of this doubling (nuts.py:97-110) -------------------------------------------------------------------
            if (!ssub) {                                   // :99 short-circuit: no draw after a stop
                const double u = draw();
                double ratio = (double)nsub / (double)n;
                ratio = ratio > 1.0 ? 1.0 : ratio;
                if (wuni(u < ratio)) sel = cand;
            }
            n += nsub;                                     // :103
            if (dir > 0) nfw += i + 1; else nbw += i + 1;
            stop = ssub || (dir > 0 ? uturn(px, pr, x, r) : uturn(x, r, px, pr));   // :105
            ++j;
            if (stop || j > a.max_depth) break;            // :89,109
            const int nd = (draw() < 0.5) ? 1 : -1;        // :91
            if (nd != dir) {                               // the other edge moves next: live <-> parked
#pragma unroll
                for (int k = 0; k < DL; ++k) {
                    double t;
                    t = x[k]; x[k] = px[k]; px[k] = t;
                    t = r[k]; r[k] = pr[k]; pr[k] = t;
                    t = g[k]; g[k] = pg[k]; pg[k] = t;
                }
                dir = nd;
            }
        }
        // ---- the selected sample by value: re-integrate from the start to leaf `sel` --------------------------------------
        double lpri1 = lpri0, llik1 = llik0, kin1 = kin_start;
        bool moved = false;
        // this tree's start comes back
        if constexpr (SMCN_WAVE_R0 == 1) {
#pragma unroll
            for (int t = 0; t < DL / 2; ++t) {
                const d2 v = r0s[t * 64];
                r[2 * t] = v.x; r[2 * t + 1] = v.y;
            }
        } else if constexpr (SMCN_WAVE_R0 == 2) {
#pragma unroll
            for (int k = 0; k < DL; ++k) r[k] = r0[k];
        } else {
            const auto ka = kargs();
            const double* const rin = ka->r;
            const bool rpm = ka->r_pm != 0;
#pragma unroll
            for (int k = 0; k < DL; ++k) r[k] = (FULL || cv[k]) ? (rpm ? rin[p * D + lane + 64 * k] : rin[cidx[k] + p]) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < DL; ++k) x[k] = x0[k];
        if (sel != 0) {
            const double e = sel > 0 ? eps : -eps, h = sel > 0 ? 0.5 * eps : -0.5 * eps;
            double ss, sl, gp[DL], gl[DL];
            model.template eval_partial_t<FULL, HAS>(x, ss, sl, gp, gl);   // (the start was finite: a leaf of weight > 0 lies behind it)
#pragma unroll
            for (int k = 0; k < DL; ++k) g[k] = grad(gp[k], gl[k], k);
            const int steps = sel > 0 ? sel : -sel;
            for (int s = 1; s < steps; ++s) bare_step(x, r, g, e, h);
            double lp;
            full_step(x, r, g, e, h, lpri1, llik1, lp, kin1);
            bool all_moved = true;
#pragma unroll
            for (int k = 0; k < DL; ++k) all_moved = all_moved && (!(FULL || cv[k]) || x[k] != x0[k]);
            moved = __builtin_amdgcn_ballot_w64(!all_moved) == 0ull;
        }
        {
            const auto ka = kargs();
            double* const xo = ka->x_new;
            double* const ro = ka->r_new;
            const bool opm = ka->r_new_pm != 0;
#pragma unroll
            for (int k = 0; k < DL; ++k) {
                if (FULL || cv[k]) {
                    xo[cidx[k] + p] = x[k];
                    if (opm) ro[p * D + lane + 64 * k] = r[k]; else ro[cidx[k] + p] = r[k];
                }
            }
            if (lane == 0) {
                if (ka->kin1) { ka->kin1[p] = kin1; ka->moved[p] = moved ? 1 : 0; }
                ka->lpri1[p] = lpri1; ka->llik1[p] = llik1;
                ka->nleap[p] = nleap; ka->depth[p] = j; ka->ndraws[p] = (int32_t)q;
                ka->flags[p] = overflow ? 1 : 0;
            }
        }
    }
}

}  // namespace smcn
