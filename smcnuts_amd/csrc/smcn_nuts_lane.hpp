// NUTS for targets a single lane can evaluate, ONE LANE PER PARTICLE, any dimension a lane can hold (PRMwCD: 13): phase 1
// of BASELINE config 4's two-phase launch -- trees of up to `jcap` doublings; longer ones are parked for nuts_fin_kernel.
//
// Replaces NUTSProposal.rvs / generate_nuts_samples / build_tree / NUTSLeapfrog / stop_criterion
// (smcnuts/proposal/nuts.py:34-175) as nuts_kernel does -- the same per-leaf state machine, the same draws at the same
// places, merges in the same order (nuts.py:134-148) -- for the mapping the arma kernel has (smcn_nuts3.hpp): 64 trees of a
// wavefront advance in lock step through the evaluation, which needs no cross-lane traffic at all (the design matrix comes
// by scalar loads), and the divergent bookkeeping is predicated per lane.
//
// What a lane kernel has to budget (the generic kernel with G = 1 spent 256 VGPRs + 256 AGPRs + 452 B of
// scratch and exposed every stack round trip):
//   * registers hold the live edge (x, r, grad) and nothing else of the tree.  A CANDIDATE is kept by reference: it is
//     either the leaf just built (the live registers) or the pending candidate of a stack level (memory); a merge that
//     keeps the older one only changes the reference, and the one copy happens where the candidate is parked at the next
//     level or accepted (nuts.py:99-101).
//   * LDS (80 doubles per lane at one wavefront per SIMD, lane-interleaved): the candidate of level 0 and the first leaves
//     of slots 1 and 2 -- 7/8 of all stack accesses.  Everything else (parked edge, selected sample, deeper levels) in an
//     HBM area of the block, lane-interleaved: lanes at the same place of their trees touch one row.
//   * the merge level m of a pass is WAVE-UNIFORM (every lane starts its merges at level 0 in the same pass; lanes only
//     drop out), so stack addresses are a scalar offset plus the lane.
//   * no work queue: lane t of the grid owns particles t, t + lanes, ...
#pragma once
#include "smcn_nuts.hpp"

namespace smcn {

template <class M, class = void>
struct model_lane_kernel { static constexpr bool value = false; };
template <class M>
struct model_lane_kernel<M, std::enable_if_t<M::LANE_KERNEL>> { static constexpr bool value = true; };

// doubles per lane of the HBM area: parked edge, selected sample, candidates of levels 1.., first leaves of slots 3..
__host__ __device__ constexpr int lane_hbm_doubles(int D) {
    return 3 * D + (2 * D + 2) + kMaxLevels * (2 * D + 3) + (kMaxLevels + 1) * 2 * D;
}
__host__ __device__ constexpr int lane_lds_doubles(int D) { return 2 * D + 2 * D; }   // candidate 0 (x, r), first leaf 1

template <class Model>
__global__ void __launch_bounds__(kNutsBlock, 1) nuts_lane_kernel(NutsArgs a) {
    constexpr int D = Model::DL;
    static_assert(Model::G == 1, "one lane per particle");
    enum { INIT = 0, LEAF = 1, DONE = 2 };
    using ldsp = __attribute__((address_space(3))) double*;
    using glbp = __attribute__((address_space(1))) double*;
    extern __shared__ double lds[];
    const int tid = (int)threadIdx.x;
    constexpr int NT = kNutsBlock;
    // ---- storage ----------------------------------------------------------------------------------------------------------
    // HBM area of the block, element k of lane t at [k][t]
    constexpr int H_EDGE = 0, H_SEL = 3 * D, H_CAND = H_SEL + 2 * D + 2, CREC = 2 * D + 3, H_FIRST = H_CAND + kMaxLevels * CREC;
    // (a block-uniform base and an index: the compiler then forms every address as scalar base + k * row + lane offset --
    //  with a per-lane pointer it kept 200+ precomputed addresses in AGPRs and paid two v_accvgpr_read per access)
    const glbp hbase = (glbp)(a.scratch + (int64_t)blockIdx.x * NT * lane_hbm_doubles(D));
    // LDS: [k][t]
    constexpr int L_CAND0 = 0, L_F1 = 2 * D;                   // (52 doubles per lane at D = 13: 104 KB; the model's row table beside them)
    constexpr int MSH = (Model::SHARED + 1) & ~1;
    double c0lp = 0.0, c0ll = 0.0, c0n = 0.0;                  // level 0's three scalars: registers
    const ldsp lbase = (ldsp)(lds + MSH);
    auto kargs = [&]() __attribute__((always_inline)) {
        using kptr = const __attribute__((address_space(4))) NutsArgs*;
        kptr kp = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        return kp;
    };
    // candidate record of level m: x[D], r[D] (k < 2 D), then lp, ll, n' (scalar 0, 1, 2)
    auto cand_st = [&](int m, int k, double v) { if (m == 0) lbase[(L_CAND0 + k) * NT + tid] = v; else hbase[(H_CAND + m * CREC + k) * NT + tid] = v; };
    auto cand_ld = [&](int m, int k) -> double { return m == 0 ? lbase[(L_CAND0 + k) * NT + tid] : hbase[(H_CAND + m * CREC + k) * NT + tid]; };
    auto cands_st = [&](int m, double vlp, double vll, double vn) {
        if (m == 0) { c0lp = vlp; c0ll = vll; c0n = vn; }
        else { hbase[(H_CAND + m * CREC + 2 * D) * NT + tid] = vlp; hbase[(H_CAND + m * CREC + 2 * D + 1) * NT + tid] = vll; hbase[(H_CAND + m * CREC + 2 * D + 2) * NT + tid] = vn; }
    };
    auto cands_ld = [&](int m, int which) -> double {
        return m == 0 ? (which == 0 ? c0lp : (which == 1 ? c0ll : c0n)) : hbase[(H_CAND + m * CREC + 2 * D + which) * NT + tid];
    };
    // first leaf of slot s (s = 1 .. kMaxLevels: nuts_kernel's FIRST + (s - 1)): x[D], r[D]
    auto first_st = [&](int s, int k, double v) {
        if (s == 1) lbase[(L_F1 + k) * NT + tid] = v; else hbase[(H_FIRST + s * 2 * D + k) * NT + tid] = v;
    };
    auto first_ld = [&](int s, int k) -> double {
        return s == 1 ? lbase[(L_F1 + k) * NT + tid] : hbase[(H_FIRST + s * 2 * D + k) * NT + tid];
    };

    Model model;
    model.init(a.mdata, 0, lds);
    const int64_t N = a.N;
    const double eps = a.eps, phi = a.phi;
    const bool taped = a.tape != nullptr;
    const int64_t lanes = (int64_t)gridDim.x * NT;

    // ---- per-lane state ---------------------------------------------------------------------------------------------------
    int64_t p = (int64_t)blockIdx.x * NT + tid;
    int phase = p < N ? (int)INIT : (int)DONE;
    double x[D], r[D], g[D];
    double logu = 0.0;
    int j = 0, i = 0, dir = 1, n = 1, nleap = 0;
    uint32_t q = 0, qbase = 0;
    double ub0 = 0.0, ub1 = 0.0;
    int64_t toff = 0, tlen = 0;
    bool overflow = false;
#pragma unroll
    for (int k = 0; k < D; ++k) { x[k] = 0.0; r[k] = 0.0; g[k] = 0.0; }

    auto refill = [&]() {
        const auto ka = kargs();
        const uint64_t seed = ka->seed;
        const u32x4 o = philox4x32_10({qbase >> 1, (uint32_t)(ka->particle_base + p), ka->iter, kStreamNuts}, (uint32_t)seed,
                                      (uint32_t)(seed >> 32));
        ub0 = u53(o.a, o.b);
        ub1 = u53(o.c, o.d);
    };
    auto draw = [&]() -> double {
        double v;
        if (taped) {
            if ((int64_t)q < tlen) v = kargs()->tape[toff + q];
            else { v = 0.5; overflow = true; }
        } else {
            if (q >= qbase + 2u) { qbase += 2u; refill(); }
            v = (q & 1u) ? ub1 : ub0;
        }
        ++q;
        return v;
    };
    auto start_particle = [&]() {
        const auto ka = kargs();
        const double* const xin = ka->x;
        const double* const rin = ka->r;
#pragma unroll
        for (int k = 0; k < D; ++k) { x[k] = xin[(int64_t)k * N + p]; r[k] = rin[(int64_t)k * N + p]; }
        q = 0; qbase = 0; overflow = false; nleap = 0;
        if (taped) { const int64_t* const to = ka->tape_off; toff = to[p]; tlen = to[p + 1] - toff; }
        else refill();
    };
    if (phase == INIT) start_particle();

    PROF_DECL;
    for (;;) {
        PROF(7);
        if (__ballot(phase != DONE) == 0ull) break;
        const bool leaf = phase == LEAF, init = phase == INIT;
        // ---- leapfrog, first half (nuts.py:169-170) ---------------------------------------------------------------------
        const double e = dir > 0 ? eps : -eps, h = dir > 0 ? 0.5 * eps : -0.5 * eps;
        if (leaf) {
#pragma unroll
            for (int k = 0; k < D; ++k) r[k] = __builtin_fma(h, g[k], r[k]);
#pragma unroll
            for (int k = 0; k < D; ++k) x[k] = __builtin_fma(e, r[k], x[k]);
        }
        // ---- target value + gradient (nuts.py:66,72,122,171): every lane ----------------------------------------------------
        double lpri, llik;
        PROF(1);
        {
            double gp[D], gl[D];
            model.eval(x, lpri, llik, gp, gl);
            double lp = lpri + phi * llik;
            const bool bad = !finite_d(lp);                 // bridgestan.py:47-49,79-80
#pragma unroll
            for (int k = 0; k < D; ++k) { const double gk = bad ? -kInf : __builtin_fma(phi, gl[k], gp[k]); if (leaf || init) g[k] = gk; }
        }
        double lp = lpri + phi * llik;
        lp = finite_d(lp) ? lp : -kInf;
        PROF(2);
        if (init) {
            // ---- nuts.py:66-87 ------------------------------------------------------------------------------------------
            double kin = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) kin = __builtin_fma(r[k], r[k], kin);
            double ex = draw();
            if (!taped) ex = -log1p(-ex);
            logu = (lp - 0.5 * kin) - ex;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                hbase[(H_EDGE + k) * NT + tid] = x[k]; hbase[(H_EDGE + D + k) * NT + tid] = r[k]; hbase[(H_EDGE + 2 * D + k) * NT + tid] = g[k];
                hbase[(H_SEL + k) * NT + tid] = x[k]; hbase[(H_SEL + D + k) * NT + tid] = r[k];
            }
            hbase[(H_SEL + 2 * D) * NT + tid] = lpri; hbase[(H_SEL + 2 * D + 1) * NT + tid] = llik;
            { const auto ka = kargs(); ka->lpri0[p] = lpri; ka->llik0[p] = llik; }
            j = 0; n = 1; i = 0;
            dir = (draw() < 0.5) ? 1 : -1;                 // nuts.py:91
            phase = LEAF;
        } else if (leaf) {
            // ---- leapfrog, second half (nuts.py:173) and leaf tests (:123-125) ---------------------------------------------
            double kin = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) { r[k] = __builtin_fma(h, g[k], r[k]); kin = __builtin_fma(r[k], r[k], kin); }
            ++nleap;
            const double joint = lp - 0.5 * kin;
            int nsub = (logu < joint) ? 1 : 0;
            bool ssub = (logu - a.delta_max) >= joint;
            int cref = -1;                                 // the candidate: -1 = this leaf (the live registers), m = level m's record
            double clp = lpri, cll = llik;
            if (j > 0 && (i & 1) == 0) {
                const int s = (i == 0) ? j : (__ffs(i) - 1);
#pragma unroll
                for (int k = 0; k < D; ++k) { first_st(s, k, x[k]); first_st(s, D + k, r[k]); }
            }
            // ---- merge completed sub-trees (nuts.py:134-148): pass m is level m for every lane still in it ------------------
            PROF(3);
            bool done = false, open = true;
            for (int m = 0; __ballot(open) != 0ull; ++m) {
                if (!open) continue;
                if (m == j) { done = true; open = false; continue; }
                if (ssub) {
                    // the stop unwinds the recursion: each ancestor for which the stopped sub-tree is the SECOND half still
                    // consumes its merge uniform
                    q += (uint32_t)__popc((unsigned)(i >> m) & ((1u << (j - m)) - 1u));
                    done = true; open = false; continue;
                }
                if (((i >> m) & 1) == 0) {                 // park the candidate as the pending first half of level m + 1
                    if (cref < 0) {
#pragma unroll
                        for (int k = 0; k < D; ++k) { cand_st(m, k, x[k]); cand_st(m, D + k, r[k]); }
                    } else {
                        // (the kept candidate lives at a lower level: copied element by element, a rolled loop -- it needs no
                        //  registers beyond one value, and happens in a quarter of the leaves)
#pragma unroll 1
                        for (int k = 0; k < 2 * D; ++k) cand_st(m, k, cand_ld(cref, k));
                    }
                    cands_st(m, clp, cll, (double)nsub);
                    open = false; continue;
                }
                const double u = draw();                   // nuts.py:142, always
                const int n1 = (int)cands_ld(m, 2);
                const int den = (n1 + nsub) > 1 ? (n1 + nsub) : 1;
                if (!(u < (double)nsub / (double)den)) { cref = m; clp = cands_ld(m, 0); cll = cands_ld(m, 1); }
                nsub += n1;                                // :146
                const int i0 = (i >> (m + 1)) << (m + 1);
                const int s = (i0 == 0) ? j : (__ffs(i0) - 1);
                double sa = 0.0, sb = 0.0;                 // nuts.py:152-160
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double fx = first_ld(s, k), fr = first_ld(s, D + k);
                    const double dx = dir > 0 ? (x[k] - fx) : (fx - x[k]);
                    const double rmn = dir > 0 ? fr : r[k];
                    const double rpl = dir > 0 ? r[k] : fr;
                    sa = __builtin_fma(dx, rmn, sa);
                    sb = __builtin_fma(dx, rpl, sb);
                }
                ssub = (sa < 0.0) || (sb < 0.0);           // :148
            }
            PROF(4);
            if (!done) {
                ++i;
            } else {
                // ---- end of this doubling (nuts.py:93-110) -----------------------------------------------------------------
                if (!ssub) {                               // :99 short-circuit: no draw after a stop
                    const double u = draw();
                    double ratio = (double)nsub / (double)n;
                    ratio = ratio > 1.0 ? 1.0 : ratio;
                    if (u < ratio) {
                        if (cref < 0) {
#pragma unroll
                            for (int k = 0; k < D; ++k) { hbase[(H_SEL + k) * NT + tid] = x[k]; hbase[(H_SEL + D + k) * NT + tid] = r[k]; }
                        } else {
#pragma unroll 1
                            for (int k = 0; k < 2 * D; ++k) hbase[(H_SEL + k) * NT + tid] = cand_ld(cref, k);
                        }
                        hbase[(H_SEL + 2 * D) * NT + tid] = clp; hbase[(H_SEL + 2 * D + 1) * NT + tid] = cll;
                    }
                }
                n += nsub;                                 // :103
                double sa = 0.0, sb = 0.0;                 // the whole trajectory's ends: live edge and parked edge (:105)
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double ox = hbase[(H_EDGE + k) * NT + tid], orr = hbase[(H_EDGE + D + k) * NT + tid];
                    const double dx = dir > 0 ? (x[k] - ox) : (ox - x[k]);
                    const double rmn = dir > 0 ? orr : r[k];
                    const double rpl = dir > 0 ? r[k] : orr;
                    sa = __builtin_fma(dx, rmn, sa);
                    sb = __builtin_fma(dx, rpl, sb);
                }
                const bool stop = ssub || (sa < 0.0) || (sb < 0.0);
                ++j;
                const auto ka = kargs();
                if (stop || j > ka->max_depth) {           // :89,109
                    double* const xo = ka->x_new;
                    double* const ro = ka->r_new;
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        xo[(int64_t)k * N + p] = hbase[(H_SEL + k) * NT + tid];
                        ro[(int64_t)k * N + p] = hbase[(H_SEL + D + k) * NT + tid];
                    }
                    ka->lpri1[p] = hbase[(H_SEL + 2 * D) * NT + tid]; ka->llik1[p] = hbase[(H_SEL + 2 * D + 1) * NT + tid];
                    ka->nleap[p] = nleap; ka->depth[p] = j; ka->ndraws[p] = (int32_t)q;
                    ka->flags[p] = overflow ? 1 : 0;
                    phase = DONE;
                } else if (ka->jcap > 0 && j == ka->jcap) {
                    // park for nuts_fin_kernel: minus edge, plus edge, selected sample, eight scalars (NutsArgs::resume)
                    double* const rec = ka->resume + p * (8 * (int64_t)D + 8);
                    const int lo = dir > 0 ? 0 : 3 * D, hi = dir > 0 ? 3 * D : 0;      // where the parked / the live edge go
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        rec[lo + k] = hbase[(H_EDGE + k) * NT + tid]; rec[lo + D + k] = hbase[(H_EDGE + D + k) * NT + tid];
                        rec[lo + 2 * D + k] = hbase[(H_EDGE + 2 * D + k) * NT + tid];
                        rec[hi + k] = x[k]; rec[hi + D + k] = r[k]; rec[hi + 2 * D + k] = g[k];
                        rec[6 * D + k] = hbase[(H_SEL + k) * NT + tid]; rec[7 * D + k] = hbase[(H_SEL + D + k) * NT + tid];
                    }
                    double* const sc = rec + 8 * D;
                    sc[0] = hbase[(H_SEL + 2 * D) * NT + tid]; sc[1] = hbase[(H_SEL + 2 * D + 1) * NT + tid]; sc[2] = logu;
                    sc[3] = (double)n; sc[4] = (double)j; sc[5] = (double)nleap; sc[6] = (double)q; sc[7] = overflow ? 1.0 : 0.0;
                    const unsigned int at = atomicAdd(ka->pend, 1u);
                    ka->pend[1 + at] = (unsigned int)p;
                    phase = DONE;
                } else {
                    const int nd = (draw() < 0.5) ? 1 : -1;   // :91
                    if (nd != dir) {                       // the other edge moves next: live <-> parked
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            const double ox = hbase[(H_EDGE + k) * NT + tid], orr = hbase[(H_EDGE + D + k) * NT + tid],
                                         og = hbase[(H_EDGE + 2 * D + k) * NT + tid];
                            hbase[(H_EDGE + k) * NT + tid] = x[k]; hbase[(H_EDGE + D + k) * NT + tid] = r[k];
                            hbase[(H_EDGE + 2 * D + k) * NT + tid] = g[k];
                            x[k] = ox; r[k] = orr; g[k] = og;
                        }
                        dir = nd;
                    }
                    i = 0;
                }
            }
        }
        PROF(5);
        // ---- the next particle of a lane that has finished one (no queue: lane t owns t, t + lanes, ...) ---------------------
        if (phase == DONE && p + lanes < N) {
            p += lanes;
            start_particle();
            phase = INIT;
        }
    }
    PROF_FLUSH(a);
}

}  // namespace smcn
