// The kernel that FINISHES the long PRMwCD trees a two-phase launch has parked (smcn_set_nuts_cap; BASELINE config 4): one
// wavefront per tree, the 13 coordinates on lanes 0..12, PrmwcdDistModel<64, ..>::eval_wave for the density.
//
// Replaces, for those trees, NUTSProposal.generate_nuts_samples / build_tree / NUTSLeapfrog / stop_criterion
// (smcnuts/proposal/nuts.py:89-175) exactly as nuts_kernel (smcn_nuts.hpp) does -- same draws at the same places, same
// merges in the same order -- and takes a tree up where nuts_kernel<.., TWO_PHASE> left it (NutsArgs::resume: the two edges,
// the selected sample, eight scalars).  With resume_in == 0 it builds whole trees from their start (smcn_set_nuts_cap's
// widen = 2: the parity tests reach this kernel on whole trees that way).
//
// Why a kernel of its own (round 5): a launch of config 4 ends with its longest tree, 1 536 leaves in here, so what counts is
// the LATENCY of one leaf.  In the generic kernel that leaf was 636 vector + 274 scalar instructions, of which the density is
// 370: the rest was the bookkeeping of a tree whose control state the compiler could not see to be wave-uniform, 64-lane
// butterflies for sums of 13 numbers, and the hybrid LDS / HBM stack.  Here
//   * the control flow is scalar by construction (conditions through a ballot, sums through v_readlane: smcn_nuts_wave.hpp);
//   * a vector is ONE double per lane, so a candidate or a first leaf is a 64-bit move, and the whole tree stack -- 11 levels
//     of (candidate x, r; first leaf x, r) -- is 16 lanes x 32 bytes per level in LDS (5.6 KB per wavefront, nothing in HBM),
//     the levels' scalars (n', the candidate's density parts) in lanes of three registers;
//   * the sums over coordinates are row sums (16 lanes), the two U-turn products sharing one through a row swap.
// Unlike the Gaussian wave kernel, candidates stay BY VALUE: replaying the selected leaf would cost a density evaluation per
// step, a third of the tree again.
#pragma once
#include "smcn_nuts_wave.hpp"

namespace smcn {

template <class M, class = void>
struct model_fin_kernel { static constexpr bool value = false; };
template <class M>
struct model_fin_kernel<M, std::enable_if_t<M::FIN_KERNEL>> { static constexpr bool value = true; };

constexpr int kFinLevels = kMaxLevels + 1;                       // slots of the first-leaf stack (slot j - 1, j <= 10) and candidate levels
__host__ __device__ constexpr int fin_lds_doubles() { return kFinLevels * 4 * 16; }   // per wavefront

#ifndef SMCN_FIN_WAVES
#define SMCN_FIN_WAVES 2     // wavefronts per SIMD the register budget is set for
#endif
template <class Model>
__global__ void __launch_bounds__(kNutsBlock, SMCN_FIN_WAVES) nuts_fin_kernel(NutsArgs a) {
    static_assert(Model::G == 64 && Model::DIST && Model::DL == 1, "one wavefront per particle, one coordinate per lane");
    using ldsd = __attribute__((address_space(3))) double*;
    extern __shared__ double lds[];
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = (int)(threadIdx.x >> 6);
    constexpr int MSH = (Model::SHARED + 1) & ~1;
    // level m: [cand x | cand r | first x | first r][16 lanes]
    const ldsd stk = (ldsd)(lds + MSH + wave * fin_lds_doubles()) + (lane & 15);
    const bool low = lane < 16;

    auto kargs = [&]() __attribute__((always_inline)) {
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
    const bool cv = lane < D;
    const int64_t cidx = (int64_t)lane * N;

    auto row0_sum = [&](double v) -> double {                    // sum over lanes 0..15 (the others hold 0), wave-uniform
        return lane_value(row_sum16(v), 0);
    };
    // nuts.py:152-160 between the trajectory ends (xm, rm) and (xp, rp)
    auto uturn = [&](double xm, double rm, double xp, double rp) -> bool {
        const double dx = xp - xm;
        const double v = row_sum16(swap16_add(dx * rm, dx * rp));     // row 0: dx . r-, row 1: dx . r+
        const double sa = lane_value(v, 0), sb = lane_value(v, 16);
        return wuni((sa < 0.0) || (sb < 0.0));
    };

    int64_t p = 0;
    uint32_t q = 0, qbase = 0;
    double ub0 = 0.0, ub1 = 0.0;
    int64_t toff = 0, tlen = 0;
    bool overflow = false;
    const bool taped = a.tape != nullptr;
    auto refill = [&]() {
        const auto ka = kargs();
        const uint64_t seed = ka->seed;
        const u32x4 o = philox4x32_10({(qbase >> 1) + (uint32_t)lane, (uint32_t)(ka->particle_base + p), ka->iter, kStreamNuts},
                                      (uint32_t)seed, (uint32_t)(seed >> 32));
        ub0 = u53(o.a, o.b);
        ub1 = u53(o.c, o.d);
    };
    auto draw = [&]() -> double {
        double v;
        if (taped) {
            if ((int64_t)q < tlen) v = wfirst(kargs()->tape[toff + q]);
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
    // value + gradient at x (bridgestan.py:47-49,79-80: a non-finite density is -inf with a gradient of -inf)
    auto density = [&](double x, double& lpri, double& llik, double& lp, double& g) {
        double xa[1] = {x}, gp[1], gl[1];
        model.eval(xa, lpri, llik, gp, gl);               // (eval_wave for the shipped shape, the generic 64-lane form otherwise)
        lp = lpri + phi * llik;
        const bool bad = !wuni(finite_d(lp));
        lp = bad ? -kInf : lp;
        g = cv ? (bad ? -kInf : __builtin_fma(phi, gl[0], gp[0])) : 0.0;
    };

    for (;;) {
        // ---- next tree: a parked one (resume_in) or a particle ------------------------------------------------------------
        unsigned int t = 0;
        if (lane == 0) t = atomicAdd(a.queue, 1u);
        t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
        const bool resumed = a.resume_in != 0;
        if (resumed) {
            if (t >= wfirst64((int64_t)a.pend[0])) break;
            p = wfirst64((int64_t)a.pend[1 + t]);
        } else {
            if ((int64_t)t >= N) break;
            p = (int64_t)t;
        }
        double x, r, g, px, pr, pg, sx, sr, x0 = 0.0;
        double slp0, slp1, logu;
        int j, n, nleap, dir;
        if (resumed) {
            const auto ka = kargs();
            const double* const rec = ka->resume + p * (8 * (int64_t)D + 8);
            const double emx = cv ? rec[lane] : 0.0, emr = cv ? rec[D + lane] : 0.0, emg = cv ? rec[2 * D + lane] : 0.0;
            const double epx = cv ? rec[3 * D + lane] : 0.0, epr = cv ? rec[4 * D + lane] : 0.0, epg = cv ? rec[5 * D + lane] : 0.0;
            sx = cv ? rec[6 * D + lane] : 0.0; sr = cv ? rec[7 * D + lane] : 0.0;
            const double* const sc = rec + 8 * D;
            slp0 = wfirst(sc[0]); slp1 = wfirst(sc[1]); logu = wfirst(sc[2]);
            n = __builtin_amdgcn_readfirstlane((int)sc[3]); j = __builtin_amdgcn_readfirstlane((int)sc[4]);
            nleap = __builtin_amdgcn_readfirstlane((int)sc[5]); q = (uint32_t)__builtin_amdgcn_readfirstlane((int)sc[6]);
            overflow = wuni(sc[7] != 0.0);
            qbase = q - (q % 128u);
            if (taped) { const int64_t* const to = ka->tape_off; toff = wfirst64(to[p]); tlen = wfirst64(to[p + 1]) - toff; }
            else refill();
            dir = (draw() < 0.5) ? 1 : -1;                 // nuts.py:91
            if (dir > 0) { x = epx; r = epr; g = epg; px = emx; pr = emr; pg = emg; }
            else { x = emx; r = emr; g = emg; px = epx; pr = epr; pg = epg; }
        } else {
            // ---- start of a tree (nuts.py:66-87) --------------------------------------------------------------------------
            const auto ka = kargs();
            x = cv ? ka->x[cidx + p] : 0.0;
            r = cv ? ka->r[cidx + p] : 0.0;
            x0 = x;
            q = 0; qbase = 0; overflow = false; nleap = 0;
            if (taped) { const int64_t* const to = ka->tape_off; toff = wfirst64(to[p]); tlen = wfirst64(to[p + 1]) - toff; }
            else refill();
            double lp;
            density(x, slp0, slp1, lp, g);
            const double kin_start = row0_sum(r * r);
            if (lane == 0) {
                ka->lpri0[p] = slp0; ka->llik0[p] = slp1;
                if (ka->kin0) ka->kin0[p] = kin_start;
            }
            double ex = draw();
            if (!taped) ex = -log1p(-ex);
            logu = (lp - 0.5 * kin_start) - ex;
            px = x; pr = r; pg = g; sx = x; sr = r;
            j = 0; n = 1;
            dir = (draw() < 0.5) ? 1 : -1;                 // nuts.py:91
        }
        int lvl_n = 0;                                     // lane m: n' of the pending first half of level m + 1 ...
        double lvl_lp = 0.0, lvl_ll = 0.0;                 // ... and the density parts of its candidate
        bool stop = false;
        for (;;) {
            // ---- one doubling: 2^j leaves in direction dir (nuts.py:93-96 -> build_tree) --------------------------------
            const double e = dir > 0 ? eps : -eps, h = dir > 0 ? 0.5 * eps : -0.5 * eps;
            int nsub = 0;
            bool ssub = false;
            double cx = 0.0, cr = 0.0, clp = 0.0, cll = 0.0;
            for (int i = 0;; ++i) {
                r = __builtin_fma(h, g, r);                // nuts.py:169-173
                x = __builtin_fma(e, r, x);
                double lpri, llik, lp;
                density(x, lpri, llik, lp, g);
                r = __builtin_fma(h, g, r);
                ++nleap;
                const double joint = lp - 0.5 * row0_sum(r * r);   // nuts.py:123-125
                nsub = wuni(logu < joint) ? 1 : 0;
                ssub = wuni((logu - a.delta_max) >= joint);
                cx = x; cr = r; clp = lpri; cll = llik;
                if (j > 0 && (i & 1) == 0) {
                    const int s = ((i == 0) ? j : __builtin_ctz((unsigned)i)) - 1;
                    if (low) { stk[(s * 4 + 2) * 16] = x; stk[(s * 4 + 3) * 16] = r; }
                }
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
                        if (low) { stk[(m * 4 + 0) * 16] = cx; stk[(m * 4 + 1) * 16] = cr; }
                        lvl_n = (lane == m) ? nsub : lvl_n;
                        lvl_lp = (lane == m) ? clp : lvl_lp;
                        lvl_ll = (lane == m) ? cll : lvl_ll;
                        break;
                    }
                    const double u = draw();               // nuts.py:142, always
                    const int n1 = __builtin_amdgcn_readlane(lvl_n, m);
                    const int den = (n1 + nsub) > 1 ? (n1 + nsub) : 1;
                    if (!wuni(u < (double)nsub / (double)den)) {
                        if (low) { cx = stk[(m * 4 + 0) * 16]; cr = stk[(m * 4 + 1) * 16]; }
                        clp = group_read<64>(lvl_lp, m); cll = group_read<64>(lvl_ll, m);
                    }
                    nsub += n1;                            // :146
                    const int i0 = (i >> (m + 1)) << (m + 1);
                    const int s = ((i0 == 0) ? j : __builtin_ctz((unsigned)i0)) - 1;
                    double fx = 0.0, fr = 0.0;
                    if (low) { fx = stk[(s * 4 + 2) * 16]; fr = stk[(s * 4 + 3) * 16]; }
                    ssub = dir > 0 ? uturn(fx, fr, x, r) : uturn(x, r, fx, fr);      // :148
                }
                if (done) break;
            }
            // ---- end of this doubling (nuts.py:97-110) -------------------------------------------------------------------
            if (!ssub) {                                   // :99 short-circuit: no draw after a stop
                const double u = draw();
                double ratio = (double)nsub / (double)n;
                ratio = ratio > 1.0 ? 1.0 : ratio;
                if (wuni(u < ratio)) { sx = cx; sr = cr; slp0 = clp; slp1 = cll; }
            }
            n += nsub;                                     // :103
            stop = ssub || (dir > 0 ? uturn(px, pr, x, r) : uturn(x, r, px, pr));   // :105
            ++j;
            if (stop || j > kargs()->max_depth) break;     // :89,109
            const int nd = (draw() < 0.5) ? 1 : -1;        // :91
            if (nd != dir) {                               // the other edge moves next: live <-> parked
                double t2;
                t2 = x; x = px; px = t2;
                t2 = r; r = pr; pr = t2;
                t2 = g; g = pg; pg = t2;
                dir = nd;
            }
        }
        {
            const auto ka = kargs();
            if (cv) { ka->x_new[cidx + p] = sx; ka->r_new[cidx + p] = sr; }
            const double* const k1 = ka->kin1;
            if (k1) {                                      // (whole trees only: a resumed tree's launch leaves these to the host path)
                const double kin_end = row0_sum(sr * sr);
                const bool every = __builtin_amdgcn_ballot_w64(cv && !(sx != x0)) == 0ull;
                if (lane == 0) { ka->kin1[p] = kin_end; ka->moved[p] = every ? 1 : 0; }
            }
            if (lane == 0) {
                ka->lpri1[p] = slp0; ka->llik1[p] = slp1;
                ka->nleap[p] = nleap; ka->depth[p] = j; ka->ndraws[p] = (int32_t)q;
                ka->flags[p] = overflow ? 1 : 0;
            }
        }
    }
}

}  // namespace smcn
